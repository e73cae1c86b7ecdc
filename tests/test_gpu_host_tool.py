"""GPU: the garlic-lod tool (C++ host adapter + ingest + C ABI) end to end on the tiny data set the
reference's prebuilt binary was run on (tests/golden/e2e, tools/make_golden_e2e.py)."""
import glob
import gzip
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E2E = os.path.join(ROOT, "tests", "golden", "e2e")
TOOL = os.path.join(ROOT, "garlic_amd", "host", "garlic-lod")


def read_rows(path):
    with gzip.open(path, "rt") as f:
        return [line.split() for line in f]


def run_tool(tmp_path, *extra, want_stderr=False):
    """--kde-subsample 0 (everyone feeds the KDE) unless the test passes its own: the default, 20 of the
    24 individuals drawn with a time seed as in the reference, would differ from run to run"""
    out = str(tmp_path / "mine")
    cmd = [TOOL, "--tped", os.path.join(E2E, "tiny.tped.gz"), "--tfam", os.path.join(E2E, "tiny.tfam"),
           "--centromere", os.path.join(E2E, "tiny.centromeres.txt"), "--error", "0.001", "--out", out]
    if "--kde-subsample" not in extra:
        cmd += ["--kde-subsample", "0"]
    r = subprocess.run(cmd + list(extra), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return (out, r.stderr) if want_stderr else out


def run_tool_tgls(tmp_path, *extra):
    """the same without --error (the likelihoods take its place)"""
    out = str(tmp_path / "mine")
    cmd = [TOOL, "--tped", os.path.join(E2E, "tiny.tped.gz"), "--tfam", os.path.join(E2E, "tiny.tfam"),
           "--centromere", os.path.join(E2E, "tiny.centromeres.txt"), "--out", out, "--kde-subsample", "0"]
    r = subprocess.run(cmd + list(extra), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


def tiny_panels():
    """independent re-parse of the tped (first non-missing allele is the counted one), monomorphic sites dropped"""
    chroms = {}
    with gzip.open(os.path.join(E2E, "tiny.tped.gz"), "rt") as f:
        for line in f:
            t = line.split()
            chroms.setdefault(t[0], []).append(t)
    cen = {l.split()[0]: (int(l.split()[1]), int(l.split()[2])) for l in open(os.path.join(E2E, "tiny.centromeres.txt"))}
    per_chr = []
    for c, rows in chroms.items():
        pos = np.array([int(float(t[3])) for t in rows], dtype=np.int32)
        geno = np.zeros((len(rows), 24), dtype=np.int16)
        freq = np.zeros(len(rows))
        for l, t in enumerate(rows):
            al = t[4:]
            one = next((x for x in al if x != "0"), "0")
            cnt = tot = 0
            for i in range(24):
                a1, a2 = al[2 * i], al[2 * i + 1]
                if a1 == "0" or a2 == "0":
                    geno[l, i] = -9
                else:
                    geno[l, i] = (a1 == one) + (a2 == one)
                for x in (a1, a2):
                    if x != "0":
                        tot += 1
                        cnt += x == one
            freq[l] = cnt / tot if tot else 0.0
        keep = (freq > 0) & (freq < 1)
        per_chr.append((geno[keep], freq[keep], pos[keep], cen[c]))
    return per_chr


def test_freq_and_raw_lod_match_reference_binary(tmp_path):
    out = run_tool(tmp_path, "--winsize", "30", "--raw-lod")
    # allele frequencies: same text
    assert gzip.open(out + ".freq.gz", "rt").read() == gzip.open(os.path.join(E2E, "ref.freq.gz"), "rt").read()
    n_tok = n_same = 0
    for ref in sorted(glob.glob(os.path.join(E2E, "ref.POP.*.raw.lod.windows.gz"))):
        mine = out + os.path.basename(ref)[3:]
        a, b = read_rows(ref), read_rows(mine)
        assert len(a) == len(b) == 24
        for ra, rb in zip(a, b):
            assert len(ra) == len(rb)
            assert [x == "NA" for x in ra] == [x == "NA" for x in rb]       # identical MISSING mask
            va = np.array([float(x) for x in ra if x != "NA"])
            vb = np.array([float(x) for x in rb if x != "NA"])
            # the prebuilt binary carries its own (older) libm; everything is printed with 6 digits
            assert np.allclose(va, vb, rtol=2e-5, atol=2e-6)
            n_tok += len(ra)
            n_same += sum(x == y for x, y in zip(ra, rb))
    assert n_same / n_tok > 0.999, (n_same, n_tok)


@pytest.mark.parametrize("tag,flags", [("refw", []), ("refp", ["--phased"])])
def test_weighted_raw_lod_matches_reference_binary(tmp_path, tag, flags):
    """--weighted end to end: genetic-map interpolation, LD weights (all individuals; hr2, or r2 from
    the TPED's allele order with --phased), wLOD -- against the raw windows the reference's prebuilt
    binary wrote for the same command (6 printed digits)."""
    out = run_tool(tmp_path, "--winsize", "30", "--raw-lod", "--weighted", "--map", os.path.join(E2E, "tiny.map"),
                   *flags)
    n_tok = n_same = 0
    for ref in sorted(glob.glob(os.path.join(E2E, tag + ".POP.*.raw.lod.windows.gz"))):
        mine = out + os.path.basename(ref)[4:]
        a, b = read_rows(ref), read_rows(mine)
        assert len(a) == len(b) == 24
        for ra, rb in zip(a, b):
            assert len(ra) == len(rb)
            assert [x == "NA" for x in ra] == [x == "NA" for x in rb]
            va = np.array([float(x) for x in ra if x != "NA"])
            vb = np.array([float(x) for x in rb if x != "NA"])
            assert np.allclose(va, vb, rtol=2e-5, atol=2e-6)
            n_tok += len(ra)
            n_same += sum(x == y for x, y in zip(ra, rb))
    assert n_tok > 100000 and n_same / n_tok > 0.999, (n_same, n_tok)


def test_tgls_raw_lod_matches_reference_binary(tmp_path):
    """--tgls --gl-type GQ end to end (readTGLSData's conversion, the filter keeping GL rows aligned,
    the TGLS kernels) against the reference binary's raw windows"""
    out = str(tmp_path / "mine")
    cmd = [TOOL, "--tped", os.path.join(E2E, "tiny.tped.gz"), "--tfam", os.path.join(E2E, "tiny.tfam"),
           "--centromere", os.path.join(E2E, "tiny.centromeres.txt"), "--out", out, "--winsize", "30", "--raw-lod",
           "--tgls", os.path.join(E2E, "tiny.tgls.gz"), "--gl-type", "GQ"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    n_tok = n_same = 0
    for ref in sorted(glob.glob(os.path.join(E2E, "reft.POP.*.raw.lod.windows.gz"))):
        mine = out + os.path.basename(ref)[4:]
        a, b = read_rows(ref), read_rows(mine)
        assert len(a) == len(b) == 24
        for ra, rb in zip(a, b):
            assert len(ra) == len(rb)
            assert [x == "NA" for x in ra] == [x == "NA" for x in rb]
            va = np.array([float(x) for x in ra if x != "NA"])
            vb = np.array([float(x) for x in rb if x != "NA"])
            assert np.allclose(va, vb, rtol=2e-5, atol=2e-6)
            n_tok += len(ra)
            n_same += sum(x == y for x, y in zip(ra, rb))
    assert n_tok > 100000 and n_same / n_tok > 0.999, (n_same, n_tok)


def test_kde_feed_matches_oracle(tmp_path):
    """<out>.<W>SNPs.lod.f64 = convertWinData2DoubleData of the scores (garlic-data.cpp:2026), bit exact."""
    out = run_tool(tmp_path, "--winsize-multi", "20", "45")
    per_chr = tiny_panels()
    for W in (20, 45):
        want = np.concatenate([
            ol.oracle_flatten(ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, 200000), W)
            for g, f, p, (cs, ce) in per_chr])
        got = np.fromfile(f"{out}.{W}SNPs.lod.f64", dtype=np.float64)
        assert ol.bits_equal(got, want), W


def test_winsize_multi_feeds_sharded_and_streamed(tmp_path):
    """--winsize-multi goes through garlic_lod_feed_multi (all sizes in one call per shard); --devices 0,0,0 merges
    three shards' feeds per size; --winsize-stream serves further window sizes on the resident panel, the loop of
    selectWinsize (garlic-roh.cpp:766-850) as the KDE's owner would drive it ('+' = previous + --auto-winsize-step)"""
    per_chr = tiny_panels()

    def want(W):
        return np.concatenate([ol.oracle_flatten(ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, 200000), W)
                               for g, f, p, (cs, ce) in per_chr])
    d = tmp_path / "sh"
    d.mkdir()
    out = run_tool(d, "--winsize-multi", "20", "45", "33", "--devices", "0,0,0")
    for W in (20, 45, 33):
        assert ol.bits_equal(np.fromfile(f"{out}.{W}SNPs.lod.f64", dtype=np.float64), want(W)), W
    out = str(tmp_path / "st")
    cmd = [TOOL, "--tped", os.path.join(E2E, "tiny.tped.gz"), "--tfam", os.path.join(E2E, "tiny.tfam"),
           "--centromere", os.path.join(E2E, "tiny.centromeres.txt"), "--error", "0.001", "--out", out, "--kde-subsample", "0",
           "--winsize", "30", "--auto-winsize", "--auto-winsize-step", "10", "--winsize-stream"]
    r = subprocess.run(cmd, input="+\n+\n25\n0\n77\n", capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    served = [l.split() for l in r.stdout.splitlines() if l.startswith("FEED ")]
    assert [int(t[1]) for t in served] == [40, 50, 25]            # 0 ends the run: 77 is never computed
    for W in (30, 40, 50, 25):
        got = np.fromfile(f"{out}.{W}SNPs.lod.f64", dtype=np.float64)
        assert ol.bits_equal(got, want(W)), W
    assert all(int(t[3]) == want(int(t[1])).shape[0] for t in served)
    assert not os.path.exists(f"{out}.77SNPs.lod.f64")


def test_kde_subsample_feed(tmp_path):
    """--kde-subsample N (selectLODCutoff -> convertSubsetWinData2DoubleData, garlic-data.cpp:2071-2150): the
    feed holds the drawn individuals only, in TFAM order; the draw is named on stderr as the reference logs
    it; a seed makes it repeatable, also across shardings and through the --raw-lod (host-side) path"""
    ids = [l.split()[1] for l in open(os.path.join(E2E, "tiny.tfam"))]
    per_chr = tiny_panels()
    feeds = []
    for k, extra in enumerate(([], ["--devices", "0,0,0"], ["--raw-lod"])):
        d = tmp_path / f"s{k}"
        d.mkdir()
        out, err = run_tool(d, "--winsize", "30", "--kde-subsample", "7", "--kde-seed", "3", *extra, want_stderr=True)
        line = next(l for l in err.splitlines() if l.startswith("Individuals used for KDE:"))
        idx = np.array([ids.index(x) for x in line.split(":")[1].split()], dtype=np.int32)
        assert idx.shape[0] == 7 and (np.diff(idx) > 0).all()
        want = np.concatenate([
            ol.oracle_flatten_subset(ol.oracle_calc_lod(g, f, p, cs, ce, 30, 0.001, 200000), 30, idx)
            for g, f, p, (cs, ce) in per_chr])
        got = np.fromfile(f"{out}.30SNPs.lod.f64", dtype=np.float64)
        assert ol.bits_equal(got, want), extra
        feeds.append((idx, got))
    assert all(np.array_equal(feeds[0][0], i) and ol.bits_equal(feeds[0][1], g) for i, g in feeds[1:])
    # the default is the reference's: 20 individuals (garlic-cli.cpp:131)
    d = tmp_path / "dflt"
    d.mkdir()
    cmd = [TOOL, "--tped", os.path.join(E2E, "tiny.tped.gz"), "--tfam", os.path.join(E2E, "tiny.tfam"),
           "--centromere", os.path.join(E2E, "tiny.centromeres.txt"), "--error", "0.001", "--out", str(d / "x"),
           "--winsize", "30"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0
    line = next(l for l in r.stderr.splitlines() if l.startswith("Individuals used for KDE:"))
    assert len(line.split(":")[1].split()) == 20


def test_sharded_run_equals_single_device(tmp_path):
    """individuals sharded over several contexts (here: all on GPU 0) -- rows gathered in TFAM order,
    LD pair counts summed over shards -- give byte-identical files, unweighted and weighted"""
    import filecmp
    weighted = ["--weighted", "--map", os.path.join(E2E, "tiny.map"), "--ld-subsample", "11", "--ld-seed", "5"]
    # with --raw-lod the full scores come back; without, only the feed thinned on the devices
    cases = ((["--winsize", "30", "--raw-lod"], []), (["--winsize", "30", "--raw-lod"], weighted),
             (["--winsize-multi", "20", "45"], []), (["--winsize", "30", "--no-kde-thinning"], weighted),
             (["--winsize", "30", "--raw-lod", "--phased"], weighted),
             # the ROH calls: every shard's segments, merged in TFAM order
             (["--winsize", "30", "--lod-cutoff", "-12", "--size-bounds", "50000", "200000"], []),
             (["--winsize", "30", "--lod-cutoff", "-4", "--size-bounds", "0.05", "0.2", "--cm"], weighted))
    for case, (common, extra) in enumerate(cases):
        outs = []
        for k, devs in enumerate(("0", "0,0", "0,0,0,0,0")):
            d = tmp_path / f"case{case}_{k}"
            d.mkdir()
            outs.append(run_tool(d, *common, *extra, "--devices", devs))
        names = sorted(os.path.basename(p)[len("mine"):] for p in glob.glob(outs[0] + "*"))
        assert any(n.endswith(".lod.f64") for n in names)
        assert ("--lod-cutoff" in common) == any(n.endswith(".roh.bed") for n in names)
        for other in outs[1:]:
            for n in names:
                assert filecmp.cmp(outs[0] + n, other + n, shallow=False), (n, other)


def test_genotype_cache_gives_identical_outputs(tmp_path):
    """runs from the 2-bit sidecar (no TPED parse; the rows go to the device as they are, 2 bits per
    genotype, also when the individuals are sharded) write byte-identical freq, raw-LOD and feed files"""
    import filecmp
    cache = str(tmp_path / "tiny.g2b")
    outs = []
    for k, devs in enumerate(("0", "0", "0,0,0")):
        d = tmp_path / f"c{k}"
        d.mkdir()
        outs.append(run_tool(d, "--winsize", "30", "--raw-lod", "--genotype-cache", cache, "--devices", devs))
    names = sorted(os.path.basename(p)[len("mine"):] for p in glob.glob(outs[0] + "*"))
    assert len(names) >= 5
    for other in outs[1:]:
        for n in names:
            assert filecmp.cmp(outs[0] + n, other + n, shallow=False), n


def test_genotype_cache_keeps_the_phase(tmp_path):
    """--phased through the sidecar: the firstCopy bits travel with the genotypes; a cache written
    without them is refused for a --phased run instead of silently giving hr2 weights"""
    import filecmp
    weighted = ["--winsize", "30", "--raw-lod", "--weighted", "--phased", "--map", os.path.join(E2E, "tiny.map")]
    cache = str(tmp_path / "tiny.g2b")
    outs = []
    for k in range(2):
        d = tmp_path / f"p{k}"
        d.mkdir()
        outs.append(run_tool(d, *weighted, "--genotype-cache", cache))
    names = sorted(os.path.basename(p)[len("mine"):] for p in glob.glob(outs[0] + "*"))
    assert len(names) >= 4
    for n in names:
        assert filecmp.cmp(outs[0] + n, outs[1] + n, shallow=False), n
    plain = str(tmp_path / "plain.g2b")
    d = tmp_path / "u"
    d.mkdir()
    run_tool(d, "--winsize", "30", "--raw-lod", "--genotype-cache", plain)
    cmd = [TOOL, "--tped", os.path.join(E2E, "tiny.tped.gz"), "--tfam", os.path.join(E2E, "tiny.tfam"),
           "--centromere", os.path.join(E2E, "tiny.centromeres.txt"), "--error", "0.001", "--out", str(d / "x"),
           *weighted, "--genotype-cache", plain]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode != 0 and "without phase" in r.stderr


def _bed(text):
    """.roh.bed -> {individual: [segment lines]} (the track lines name the individuals)"""
    out, cur = {}, None
    for line in text.splitlines():
        if line.startswith("track"):
            cur = line.split('"')[1].split()[1]
            out[cur] = []
        else:
            out[cur].append(line)
    return out


@pytest.mark.parametrize("tag,flags", [
    ("ref", ["--lod-cutoff", "-12", "--size-bounds", "50000", "200000"]),
    ("refw", ["--weighted", "--map", os.path.join(E2E, "tiny.map"), "--ld-subsample", "0", "--cm", "--lod-cutoff", "-4",
              "--size-bounds", "0.05", "0.2"]),
    ("reft", ["--tgls", os.path.join(E2E, "tiny.tgls.gz"), "--gl-type", "GQ", "--lod-cutoff", "-11", "--size-bounds", "50000", "200000"])])
def test_roh_calls_match_reference_binary(tmp_path, tag, flags):
    """--lod-cutoff / --size-bounds: calcLODWindows + assembleROHWindows + writeROHData of the prebuilt binary (the
    .roh.bed in tests/golden/e2e) against the tool, whose device goes from the genotypes to the segments without scores
    or counts: every individual's chromosome / start / stop / size class / size / colour lines, in bp and in cM, and from
    per-genotype likelihoods"""
    if tag == "reft":
        out = run_tool_tgls(tmp_path, "--winsize", "30", *flags)
    else:
        out = run_tool(tmp_path, "--winsize", "30", *flags)
    ref_text = gzip.open(os.path.join(E2E, tag + ".roh.bed.gz"), "rt").read()
    mine_text = open(out + ".roh.bed").read()
    ref = _bed(ref_text)
    mine = _bed(mine_text)
    assert list(mine) == list(ref) and len(ref) == 24
    assert sum(len(v) for v in ref.values()) > 100
    assert mine == ref
    assert mine_text == ref_text          # track lines (Ind / Pop / version text, order) included: the whole file
