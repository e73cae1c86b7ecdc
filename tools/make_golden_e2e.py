#!/usr/bin/env python3
"""End-to-end text fixture: a tiny TPED/TFAM/centromere data set and what
the reference's PREBUILT binary (/root/reference/bin/linux/garlic, v1.1.6a, static) writes for it:
the allele-frequency file and the raw LOD windows (--raw-lod).  Build container only.
The binary embeds its own libm and prints 6 significant digits, so this pins ingest, plumbing and
formats -- the bit-level pin is tools/make_golden.py."""
import glob
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "e2e")
BIN = "/root/reference/bin/linux/garlic"


def main():
    rng = np.random.default_rng(20260105)
    nind, W = 24, 30
    os.makedirs(OUT, exist_ok=True)
    lines = []
    cen = []
    alleles = "ACGT"
    for c, n in ((1, 2500), (2, 2000), (7, 1500)):
        pos = np.cumsum(rng.integers(200, 4000, size=n))
        pos[n // 3:] += 250000                       # one > max_gap hole
        k0 = n // 2
        cen.append((c, int(pos[k0]) + 1, int(pos[k0]) + 40000))
        pos[k0 + 1:] += 50000                        # centromere gap with no SNP inside
        freq = rng.uniform(0.05, 0.95, size=n)
        for l in range(n):
            a, b = rng.choice(4, size=2, replace=False)
            A, B = alleles[a], alleles[b]
            g = []
            for i in range(nind):
                if rng.random() < 0.02:
                    g += ["0", "0"]
                else:
                    g += [A if rng.random() < freq[l] else B, A if rng.random() < freq[l] else B]
            lines.append(f"{c} rs{c}_{l} 0 {int(pos[l])} " + " ".join(g))
    with open(os.path.join(OUT, "tiny.tped.gz"), "wb") as raw, \
            gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0) as f:   # mtime 0: same bytes every run
        f.write(("\n".join(lines) + "\n").encode())
    with open(os.path.join(OUT, "tiny.tfam"), "w") as f:
        for i in range(nind):
            f.write(f"POP ind{i} 0 0 0 -9\n")
    with open(os.path.join(OUT, "tiny.centromeres.txt"), "w") as f:
        for c, s, e in cen:
            f.write(f"{c} {s} {e}\n")
    tmp = tempfile.mkdtemp()
    cmd = [BIN, "--tped", os.path.join(OUT, "tiny.tped.gz"), "--tfam", os.path.join(OUT, "tiny.tfam"),
           "--centromere", os.path.join(OUT, "tiny.centromeres.txt"), "--error", "0.001", "--winsize", str(W),
           "--raw-lod", "--kde-subsample", "0", "--out", os.path.join(tmp, "ref")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    print(r.stdout[-400:], r.stderr[-800:])
    for p in glob.glob(os.path.join(tmp, "ref*")):
        print(" ", os.path.basename(p), os.path.getsize(p))
    shutil.copy(os.path.join(tmp, "ref.freq.gz"), os.path.join(OUT, "ref.freq.gz"))
    for p in glob.glob(os.path.join(tmp, "ref.*.raw.lod.windows.gz")):
        shutil.copy(p, os.path.join(OUT, os.path.basename(p)))
    # --weighted: genetic map (chr snpid gpos ppos, ~1 cM/Mb with jitter, every 3rd SNP so that the
    # rest is interpolated), LD weights from all individuals (--ld-subsample 0: no random draw)
    rng2 = np.random.default_rng(20260106)
    with open(os.path.join(OUT, "tiny.map"), "w") as f:
        for line in lines[::3]:
            t = line.split()[:4]
            f.write(f"{t[0]} {t[1]} {int(t[3]) * 1e-6 * rng2.uniform(0.9, 1.1):.8f} {t[3]}\n")
    cmdw = cmd[:-1] + [os.path.join(tmp, "refw"), "--weighted", "--map", os.path.join(OUT, "tiny.map"),
                       "--threads", "8"]
    r = subprocess.run(cmdw, capture_output=True, text=True)
    print(r.stdout[-400:], r.stderr[-800:])
    for p in glob.glob(os.path.join(tmp, "refw.*.raw.lod.windows.gz")):
        print(" ", os.path.basename(p), os.path.getsize(p))
        shutil.copy(p, os.path.join(OUT, os.path.basename(p)))
    # --weighted --phased: r2 from the allele order in the TPED (first allele of each pair)
    cmdp = cmd[:-1] + [os.path.join(tmp, "refp"), "--weighted", "--phased", "--map", os.path.join(OUT, "tiny.map"),
                       "--threads", "8"]
    r = subprocess.run(cmdp, capture_output=True, text=True)
    print(r.stdout[-400:], r.stderr[-800:])
    for p in glob.glob(os.path.join(tmp, "refp.*.raw.lod.windows.gz")):
        print(" ", os.path.basename(p), os.path.getsize(p))
        shutil.copy(p, os.path.join(OUT, os.path.basename(p)))
    # --tgls: per-genotype GQ (chr snpid gpos ppos, then one integer per individual), no --error
    rng3 = np.random.default_rng(20260107)
    with open(os.path.join(OUT, "tiny.tgls.gz"), "wb") as raw, \
            gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0) as f:
        for line in lines:
            t = line.split()[:4]
            f.write((" ".join(t) + " " + " ".join(str(int(q)) for q in rng3.integers(3, 61, size=nind)) + "\n").encode())
    cmdt = [x for x in cmd[:-1] if x not in ("--error", "0.001")] + \
           [os.path.join(tmp, "reft"), "--tgls", os.path.join(OUT, "tiny.tgls.gz"), "--gl-type", "GQ"]
    r = subprocess.run(cmdt, capture_output=True, text=True)
    print(r.stdout[-400:], r.stderr[-800:])
    for p in glob.glob(os.path.join(tmp, "reft.*.raw.lod.windows.gz")):
        print(" ", os.path.basename(p), os.path.getsize(p))
        shutil.copy(p, os.path.join(OUT, os.path.basename(p)))
    # the final pass: --lod-cutoff (no KDE) and --size-bounds (no GMM) leave calcLODWindows + assembleROHWindows +
    # writeROHData -- the ROH calls in bp from the unweighted scores, in cM (--cm) from the weighted ones, in bp from the
    # scores with per-genotype likelihoods
    cmdr = [x for x in cmd[:-1] if x != "--raw-lod"] + [os.path.join(tmp, "ref"), "--lod-cutoff", "-12", "--size-bounds", "50000", "200000"]
    cmdrw = [x for x in cmdw if x != "--raw-lod"] + ["--cm", "--lod-cutoff", "-4", "--size-bounds", "0.05", "0.2"]
    cmdrt = [x for x in cmdt if x != "--raw-lod"] + ["--lod-cutoff", "-11", "--size-bounds", "50000", "200000"]
    for c, tag in ((cmdr, "ref"), (cmdrw, "refw"), (cmdrt, "reft")):
        r = subprocess.run(c, capture_output=True, text=True)
        print(r.stdout[-400:], r.stderr[-800:])
        with open(os.path.join(tmp, tag + ".roh.bed"), "rb") as src, open(os.path.join(OUT, tag + ".roh.bed.gz"), "wb") as raw, \
                gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0) as f:
            f.write(src.read())
    with open(os.path.join(OUT, "COMMAND.txt"), "w") as f:
        f.write("garlic v1.1.6a prebuilt binary:\n")
        for c in (cmd, cmdw, cmdp, cmdt, cmdr, cmdrw, cmdrt):
            f.write(" ".join(os.path.basename(x) if x.startswith("/") else x for x in c) + "\n")
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
