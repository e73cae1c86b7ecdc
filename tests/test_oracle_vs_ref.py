"""CPU, build container only: the oracle against the real reference (oracle/_ref) on random panels.
Skipped where oracle/_ref was never built (it cannot be rebuilt without /root/reference)."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref/libgarlic_ref.so not built")


def test_lod_inGap_decay_scalars():
    rng = np.random.default_rng(0)
    o, r = ol.oracle(), ol.ref()
    for _ in range(2000):
        g = int(rng.choice([0, 1, 2, -9, 3]))
        f = float(rng.choice([0.0, 1.0, rng.uniform(0, 1)]))
        e = float(rng.choice([1e-16, 1e-3, rng.uniform(0, 1), 1.0]))
        assert np.float64(o.oracle_lod(g, f, e)).view(np.uint64) == np.float64(r.ref_lod(g, f, e)).view(np.uint64)
        q = [int(x) for x in rng.integers(0, 50, size=4)]
        assert o.oracle_in_gap(*q) == r.ref_inGap(*q)
        iv = float(rng.uniform(0, 1e6))
        assert o.oracle_nomut(7.0, 1e-9, iv) == r.ref_nomut(7.0, 1e-9, iv)
        assert o.oracle_norec(7.0, iv * 1e-6) == r.ref_norec(7.0, iv * 1e-6)


def test_unweighted_random_panels():
    rng = np.random.default_rng(1)
    total = 0
    for _ in range(120):
        nloci, nind, W = int(rng.integers(1, 400)), int(rng.integers(1, 9)), int(rng.integers(2, 70))
        mg = int(rng.choice([200000, 3000, 50000]))
        geno, freq, pos, cS, cE = ol.random_panel(rng, nloci, nind, max_gap=mg)
        known = bool(rng.integers(0, 2))
        if not known:
            cS = cE = 0
        gl = rng.choice([1e-16, 1e-3, 0.01, 0.5, 1.0, 10 ** -2.7], size=geno.shape) if rng.integers(0, 2) else None
        a = ol.oracle_calc_lod(geno, freq, pos, cS, cE, W, 0.001, mg, gl=gl)
        b = ol.ref_calc_lod(geno, freq, pos, cS, cE, W, 0.001, mg, gl=gl, centro_known=known)
        assert ol.bits_equal(a, b)
        assert ((b != ol.MISSING) == ol.oracle_mask(pos, cS, cE, W, mg)[None, :].astype(bool)).all()
        total += a.size
    assert total > 50000


def test_wlod_random_panels():
    rng = np.random.default_rng(2)
    for _ in range(40):
        nloci, nind, W = int(rng.integers(2, 300)), int(rng.integers(2, 9)), int(rng.integers(2, 40))
        geno, freq, pos, cS, cE = ol.random_panel(rng, nloci, nind)
        gpos = pos * 1e-6 * rng.uniform(0.8, 1.2)
        hom, ld = ol.ref_hr2_ld(geno, W, threads=int(rng.integers(1, 4)))
        assert ol.bits_equal(hom, ol.oracle_geno_freq(geno))
        assert ol.bits_equal(ld, ol.oracle_hr2_ld(geno, W))
        ld = np.where(np.isfinite(ld) & (ld > 0), ld, 1.0)
        gl = rng.choice([1e-16, 1e-3, 0.5, 1.0], size=geno.shape) if rng.integers(0, 2) else None
        a = ol.oracle_calc_wlod(geno, freq, pos, gpos, ld, cS, cE, W, 0.001, 200000, 1e-9, 7, gl=gl,
                                threads=int(rng.integers(1, 6)))
        b = ol.ref_calc_wlod(geno, freq, pos, gpos, ld, cS, cE, W, 0.001, 200000, 1e-9, 7, gl=gl,
                             threads=int(rng.integers(1, 6)))
        assert ol.bits_equal(a, b)


def test_ld_subsample_index():
    """hr2 over an injected --ld-subsample index (garlic-data.cpp:361-364, 562-566): oracle == reference"""
    rng = np.random.default_rng(9)
    for _ in range(25):
        nloci, nind, W = int(rng.integers(5, 200)), int(rng.integers(3, 40)), int(rng.integers(2, 30))
        geno = ol.random_panel(rng, nloci, nind, miss=float(rng.choice([0.0, 0.05, 0.4])))[0]
        idx = np.sort(rng.choice(nind, size=int(rng.integers(1, nind + 1)), replace=False)).astype(np.int32)
        hom, ld = ol.ref_hr2_ld(geno, W, idx=idx, threads=int(rng.integers(1, 4)))
        mine = ol.oracle_hr2_ld(geno, W, idx=idx)
        nan = np.isnan(ld)
        assert np.array_equal(nan, np.isnan(mine)) and ol.bits_equal(ld[~nan], mine[~nan])



def test_phased_r2_ld():
    """--phased LD weights (calcR2LD / r2, garlic-data.cpp:426-535, 585-617): oracle == reference"""
    rng = np.random.default_rng(31)
    for _ in range(25):
        nloci, nind, W = int(rng.integers(5, 200)), int(rng.integers(3, 40)), int(rng.integers(2, 30))
        geno, freq = ol.random_panel(rng, nloci, nind, miss=float(rng.choice([0.0, 0.05, 0.4])))[:2]
        freq = freq.copy()
        freq[rng.random(nloci) < 0.1] = rng.choice([0.0, 1.0])       # monomorphic by frequency: r2 = 0
        fc = rng.integers(0, 2, size=geno.shape).astype(np.uint8)
        idx = None
        if rng.integers(0, 2):
            idx = np.sort(rng.choice(nind, size=int(rng.integers(1, nind + 1)), replace=False)).astype(np.int32)
        ld = ol.ref_r2_ld(geno, fc, freq, W, idx=idx, threads=int(rng.integers(1, 4)))
        mine = ol.oracle_r2_ld(geno, fc, freq, W, idx=idx)
        nan = np.isnan(ld)
        assert np.array_equal(nan, np.isnan(mine)) and ol.bits_equal(ld[~nan], mine[~nan])


def _segments_from_coverage(inwin, pos, cS, cE, W, max_gap, overlap_frac):
    """second half of assembleROHWindows (garlic-roh.cpp:456-533), applied to coverage counts the
    caller supplies -- only here, to pin those counts through the segments the reference reports"""
    thr = overlap_frac * W
    thr = thr if thr >= 1 else 1
    thr = thr if thr <= W else W
    out = []
    n = len(pos)
    start = start_idx = -1
    for w in range(n):
        hit = inwin[w] >= thr
        if start < 0 and hit:
            start, start_idx = int(pos[w]), w
        elif hit and (int(pos[w]) - int(pos[w - 1]) > max_gap or ol.oracle().oracle_in_gap(int(pos[w - 1]), int(pos[w]), cS, cE)):
            if (w - 1) - start_idx + 1 >= thr:
                out.append((float(start), float(pos[w - 1])))
            start, start_idx = int(pos[w]), w
        elif start > 0 and not hit:
            if (w - 1) - start_idx + 1 >= thr:
                out.append((float(start), float(pos[w - 1])))
            start = start_idx = -1
        elif start > 0 and w + 1 >= n:
            if w - start_idx + 1 >= thr:
                out.append((float(start), float(pos[w])))
            start = start_idx = -1
    return out


def test_subset_flatten():
    """convertSubsetWinData2DoubleData (garlic-data.cpp:2071-2150) needs GSL for its draw and cannot be
    linked here; its two loops are convertWinData2DoubleData's with data[randInd[ind]] for data[ind], so
    the oracle's restatement (drawn individuals injected) must equal the REAL convertWinData2DoubleData
    on the rows win[randInd] -- in the draw's order, ascending as gsl_ran_choose leaves it, or any other"""
    rng = np.random.default_rng(17)
    for _ in range(60):
        nloci, nind = int(rng.integers(1, 300)), int(rng.integers(1, 40))
        win = rng.normal(size=(nind, nloci))
        win[rng.random(win.shape) < 0.2] = ol.MISSING
        win[rng.random(win.shape) < 0.02] = np.nan
        step = int(rng.choice([1, 7, 30, nloci + 5]))
        k = int(rng.integers(1, nind + 1))
        idx = rng.choice(nind, size=k, replace=False).astype(np.int32)
        for order in (np.sort(idx), idx):
            want = ol.ref_flatten(np.ascontiguousarray(win[order]), step)
            assert ol.bits_equal(ol.oracle_flatten_subset(win, step, order), want)
        assert ol.bits_equal(ol.oracle_flatten_subset(win, step, np.arange(nind)), ol.ref_flatten(win, step))


def test_roh_coverage_through_the_reference_segments():
    """oracle_roh_coverage restates the inWin[] loop in the middle of assembleROHWindows
    (garlic-roh.cpp:446-454), which the reference does not expose.  Pinned here through what it feeds:
    coverage counts -> ROH segments (restated above) must equal the segments the real
    assembleROHWindows reports, for thresholds from 1 SNP to the whole window."""
    rng = np.random.default_rng(77)
    checked = 0
    for _ in range(12):
        nloci, nind, W = int(rng.integers(60, 400)), int(rng.integers(1, 6)), int(rng.integers(2, 25))
        geno, freq, pos, cS, cE = ol.random_panel(rng, nloci, nind)
        mg = 200000
        win = ol.oracle_calc_lod(geno, freq, pos, cS, cE, W, 0.001, mg)
        cutoff = float(np.quantile(win[win != ol.MISSING], rng.uniform(0.3, 0.8))) if (win != ol.MISSING).any() else 0.0
        cov = ol.oracle_roh_coverage(win, W, cutoff)
        for frac in (1e-9, 0.25, 0.5, 0.77, 1.0):
            ref = ol.ref_assemble_roh(win, pos, cS, cE, cutoff, W, mg, frac)
            mine = [(i, a, b) for i in range(nind)
                    for a, b in _segments_from_coverage(cov[i], pos, cS, cE, W, mg, frac)]
            assert mine == ref, (nloci, nind, W, frac)
            # the C restatement the GPU tests use (SNP indices): the same segments
            segs = ol.oracle_roh_segments(cov, pos, cS, cE, W, mg, frac)
            assert [(i, float(pos[a]), float(pos[b])) for i, a, b in segs] == ref, (nloci, nind, W, frac)
            checked += len(ref)
    assert checked > 200


def test_roh_segments_at_the_edges_of_the_state_machine():
    """oracle_roh_segments against the real assembleROHWindows where its four branches meet: a qualifying last SNP
    (a segment that begins there is never reported), a break right before the last SNP, breaks inside a stretch of
    qualifying SNPs, a chromosome of one SNP, every SNP qualifying, thresholds of one SNP and of the whole window"""
    rng = np.random.default_rng(5)
    checked = 0
    for trial in range(40):
        nloci, nind, W = int(rng.integers(1, 70)), int(rng.integers(1, 4)), int(rng.integers(2, 9))
        pos = np.cumsum(rng.integers(1, 3000, size=nloci)).astype(np.int32) + 1000
        mg = 5000
        for k in rng.integers(1, max(2, nloci), size=3):        # a few breaks, one of them often before the last SNP
            if 0 < k < nloci:
                pos[k:] += mg + 1
        if nloci > 1 and trial % 3 == 0:
            pos[-1:] += mg + 1
        cS, cE = (int(pos[nloci // 2]) + 1, int(pos[nloci // 2]) + 2) if trial % 2 else (0, 0)
        win = np.where(rng.random((nind, nloci)) < 0.7, 5.0, -5.0)
        if trial % 4 == 0:
            win[:] = 5.0
        win[:, max(0, nloci - W + 1):] = ol.MISSING                # windows that do not fit hold no score
        if nloci >= W:
            win[:, nloci - W] = 5.0                                # the last window qualifies: the last SNP is covered
        cov = ol.oracle_roh_coverage(win, W, 0.0)
        for frac in (1e-9, 0.3, 1.0):
            ref = ol.ref_assemble_roh(win, pos, cS, cE, 0.0, W, mg, frac)
            segs = ol.oracle_roh_segments(cov, pos, cS, cE, W, mg, frac)
            assert [(i, float(pos[a]), float(pos[b])) for i, a, b in segs] == ref, (trial, nloci, W, frac)
            checked += len(ref)
    assert checked > 100


def test_roh_segments_of_a_chromosome_that_starts_at_position_zero():
    """a 0-based map: the reference tells 'a segment is open' by its first POSITION being > 0 and 'none' by < 0
    (garlic-roh.cpp:456, 493, 514), so a stretch opened at SNP 0 is neither -- it closes only at a covered SNP behind a
    break and is reported from SNP 0 whatever lies between.  The oracle's walk against the real assembleROHWindows there:
    SNP 0 covered or not, with and without a covered break further on"""
    rng = np.random.default_rng(11)
    checked = wedged = 0
    for trial in range(60):
        nloci, nind, W = int(rng.integers(1, 90)), int(rng.integers(1, 4)), int(rng.integers(2, 9))
        pos = np.cumsum(rng.integers(1, 3000, size=nloci)).astype(np.int32)
        pos -= pos[0]
        mg = 5000
        if trial % 3:
            for k in rng.integers(1, max(2, nloci), size=2):
                if 0 < k < nloci:
                    pos[k:] += mg + 1
        cS, cE = (int(pos[nloci // 2]) + 1, int(pos[nloci // 2]) + 2) if trial % 2 else (0, 0)
        win = np.where(rng.random((nind, nloci)) < 0.6, 5.0, -5.0)
        if trial % 4 == 0:
            win[:, 0] = 5.0
        win[:, max(0, nloci - W + 1):] = ol.MISSING
        cov = ol.oracle_roh_coverage(win, W, 0.0)
        for frac in (1e-9, 0.3, 1.0):
            ref = ol.ref_assemble_roh(win, pos, cS, cE, 0.0, W, mg, frac)
            segs = ol.oracle_roh_segments(cov, pos, cS, cE, W, mg, frac)
            assert [(i, float(pos[a]), float(pos[b])) for i, a, b in segs] == ref, (trial, nloci, W, frac)
            checked += len(ref)
            wedged += sum(1 for i, a, b in segs if a == 0)
    assert checked > 100 and wedged > 10


def test_genetic_map_interpolation_of_the_host_adapter(tmp_path):
    """garlic_amd/host's loadMapScaffold + interpolateGeneticmap (what --weighted feeds wLOD with) against
    the reference's own functions (garlic-data.cpp:702-757 through oracle/_ref): every genetic position
    bit for bit -- exact scaffold hits, interpolated sites, the scaffold's first and last site"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "map_unit")
    libdir = os.path.join(root, "garlic_amd")
    cc = subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-o", exe, os.path.join(root, "tests", "host_unit", "map_unit.cpp"),
                         "-L" + libdir, "-lgarlic_host", "-lgarlic_hip", "-lz", "-ldl", "-Wl,-rpath," + libdir],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-3000:]
    rng = np.random.default_rng(8)
    for trial in range(5):
        n = int(rng.integers(2, 400))
        pp = np.cumsum(rng.integers(1, 5000, size=n)).astype(np.int64) + 1000
        gp = np.cumsum(rng.uniform(0.0, 0.01, size=n))
        mapfile = str(tmp_path / f"m{trial}.map")
        with open(mapfile, "w") as f:
            for a, b in zip(pp, gp):
                f.write(f"chrT rs{a} {b:.10f} {a}\n")
        q = np.unique(np.concatenate([rng.integers(pp[0], pp[-1] + 1, size=300), pp[rng.integers(0, n, size=20)],
                                      [pp[0], pp[-1]]]))
        r = subprocess.run([exe, os.path.join(root, "oracle", "_ref", "libgarlic_ref.so"), mapfile, "chrT"] + [str(int(x)) for x in q],
                           capture_output=True, text=True)
        assert r.returncode == 0 and "map_unit ok" in r.stdout, (trial, (r.stdout + r.stderr)[-2000:])


@pytest.mark.parametrize("gl_type", ["GQ", "GL", "PL"])
def test_tgls_reader_of_the_host_adapter(tmp_path, gl_type):
    """garlic_amd/host's readTGLSData, as doubles and as dictionary codes, against the reference's reader:
    the conversions of garlic-data.cpp:1557-1576 incl. the clamps at 1e-16 and 1, bit for bit"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "tgls_unit")
    libdir = os.path.join(root, "garlic_amd")
    cc = subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-o", exe, os.path.join(root, "tests", "host_unit", "tgls_unit.cpp"),
                         "-L" + libdir, "-lgarlic_host", "-lgarlic_hip", "-lz", "-ldl", "-Wl,-rpath," + libdir],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-3000:]
    rng = np.random.default_rng({"GQ": 1, "GL": 2, "PL": 3}[gl_type])
    nloci, nind = 300, 17
    if gl_type == "GQ":
        vals = rng.choice(np.concatenate([np.arange(0, 100), [150, 1000]]), size=(nloci, nind)).astype(float)
    elif gl_type == "GL":
        vals = rng.choice([-0.0004, -0.004, -0.3, -1.0, -3.0, -12.0, 0.0, 0.1], size=(nloci, nind))
    else:
        vals = rng.choice([0.004, 0.04, 3.0, 10.0, 30.0, 120.0, 0.0, -1.0], size=(nloci, nind))
    path = str(tmp_path / "x.tgls")
    with open(path, "w") as f:
        for l in range(nloci):
            f.write(f"1 rs{l} 0 {l + 1} " + " ".join(repr(float(v)) if gl_type != "GQ" else str(int(v)) for v in vals[l]) + "\n")
    r = subprocess.run([exe, os.path.join(root, "oracle", "_ref", "libgarlic_ref.so"), path, gl_type, str(nloci), str(nind)],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "tgls_unit ok" in r.stdout, (r.stdout + r.stderr)[-2000:]
