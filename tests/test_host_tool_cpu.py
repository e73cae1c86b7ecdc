"""CPU: garlic-lod's flag validation mirrors the reference's Phase-I validators (src/garlic-cli.cpp:240-462)
and the tool fails loudly when there is no GPU (no CPU fallback)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E2E = os.path.join(ROOT, "tests", "golden", "e2e")
TOOL = os.path.join(ROOT, "garlic_amd", "host", "garlic-lod")
BASE = ["--tped", os.path.join(E2E, "tiny.tped.gz"), "--tfam", os.path.join(E2E, "tiny.tfam")]


def run(*args):
    return subprocess.run([TOOL, *args], capture_output=True, text=True)


@pytest.mark.parametrize("args,msg", [
    (BASE + ["--error", "0.001", "--winsize", "30"], "--build or --centromere"),
    (BASE + ["--build", "hg19", "--winsize", "30"], "error rate must be > 0 and < 1"),
    (BASE + ["--build", "hg19", "--error", "0.001", "--winsize", "1"], "window size must be > 1"),
    (BASE + ["--build", "hg19", "--error", "0.001", "--winsize", "30", "--overlap-frac", "1.5"], "Overlap fraction"),
    (BASE + ["--build", "hg19", "--tgls", "x.tgls", "--winsize", "30"], "GQ/GL/PL"),
    (BASE + ["--build", "hg19", "--error", "0.001", "--winsize", "30", "--weighted"], "--map"),
])
def test_validators(args, msg):
    r = run(*args)
    assert r.returncode != 0 and msg in r.stderr


def test_no_gpu_is_a_loud_failure(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = run(*BASE, "--centromere", os.path.join(E2E, "tiny.centromeres.txt"), "--error", "0.001",
            "--winsize", "30", "--out", str(tmp_path / "x"))
    assert r.returncode != 0
    assert "Loaded 6000 loci x 24 individuals (3 chromosomes)" in r.stderr   # ingest ran
    assert "no HIP device" in r.stderr or "garlic_ctx_create" in r.stderr


def test_genotype_cache_round_trip_without_gpu(tmp_path):
    """--genotype-cache: the first run parses the TPED and writes the 2-bit sidecar, the second loads it
    (same loci x individuals x chromosomes); ingest is host code, so this runs without a GPU too"""
    cache = str(tmp_path / "tiny.g2b")
    args = [*BASE, "--centromere", os.path.join(E2E, "tiny.centromeres.txt"), "--error", "0.001",
            "--winsize", "30", "--out", str(tmp_path / "x"), "--genotype-cache", cache]
    r1 = run(*args)
    assert "Wrote genotype cache" in r1.stderr and os.path.getsize(cache) > 6000 * 6
    r2 = run(*args)
    assert "Loaded genotype cache" in r2.stderr and "Wrote genotype cache" not in r2.stderr
    assert "Loaded 6000 loci x 24 individuals (3 chromosomes)" in r2.stderr
    # a truncated cache is refused, not half-read
    with open(cache, "r+b") as f:
        f.truncate(os.path.getsize(cache) // 2)
    r3 = run(*args)
    assert r3.returncode != 0 and "truncated" in r3.stderr


def test_compact_host_containers(tmp_path):
    """tests/host_unit/host_unit.cpp (compiled here, ASan + UBSan): the packed genotype rows of the cache and
    the dictionary-coded likelihoods hold exactly what the reference-shaped containers hold, before and
    after the monomorphic-site filter; a cache rewritten from packed rows is byte-identical"""
    exe = str(tmp_path / "host_unit")
    src = os.path.join(ROOT, "tests", "host_unit", "host_unit.cpp")
    libdir = os.path.join(ROOT, "garlic_amd")
    cc = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined",
                         "-fno-omit-frame-pointer", "-o", exe, src, os.path.join(libdir, "host", "garlic_host.cpp"),
                         "-L" + libdir, "-lgarlic_hip", "-lz", "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-3000:]
    r = subprocess.run([exe, os.path.join(E2E, "tiny.tped.gz"), os.path.join(E2E, "tiny.tgls.gz"), "GQ", str(tmp_path)],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "host_unit ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_resample_is_the_gsl_mt19937_stream(tmp_path):
    """--resample N (garlic-data.cpp:16-20, 142-148): per SNP, in file order, N draws of one mt19937 stream
    (GSL's default generator, gsl_rng_uniform = 32 bits / 2^32); with --resample-seed the draw is repeatable and
    equals the textbook generator seeded alike (numpy's legacy RandomState = init_genrand, as GSL).  Ingest is
    host code: the .freq.gz is written before the tool asks for a GPU."""
    import gzip
    import numpy as np
    outs = {}
    for tag, extra in (("plain", []), ("r7", ["--resample", "50", "--resample-seed", "7"]),
                       ("r7b", ["--resample", "50", "--resample-seed", "7"]), ("r8", ["--resample", "50", "--resample-seed", "8"])):
        out = str(tmp_path / tag)
        run(*BASE, "--centromere", os.path.join(E2E, "tiny.centromeres.txt"), "--error", "0.001", "--winsize", "30",
            "--out", out, *extra)
        outs[tag] = np.array([float(l.split()[-1]) for l in list(gzip.open(out + ".freq.gz", "rt"))[1:]])
    assert np.array_equal(outs["r7"], outs["r7b"]) and not np.array_equal(outs["r7"], outs["r8"])
    assert np.allclose(outs["r7"] * 50, np.round(outs["r7"] * 50))
    raw = np.random.RandomState(7).randint(0, 2 ** 32, size=outs["plain"].shape[0] * 50, dtype=np.uint64) / 4294967296.0
    want = np.array([np.count_nonzero(raw[50 * k: 50 * k + 50] <= f) / 50.0 for k, f in enumerate(outs["plain"])])
    # (.freq.gz prints 6 digits: a draw within 1e-6 of the frequency may fall on the other side)
    assert np.count_nonzero(want != outs["r7"]) <= 2
