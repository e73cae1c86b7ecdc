#!/usr/bin/env python3
"""One-off soak on the GPU box: many random panels through every variant of the path, each result
against the CPU oracle bit for bit.  Not part of the test suite (minutes, not seconds):

    python tools/soak.py [--trials 150] [--seed 1]

Prints one line per failure and a summary; exit code 1 if anything differed."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def same(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    if a.shape != b.shape or not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    ok = ~np.isnan(a)
    return np.array_equal(a[ok].view(np.uint64), b[ok].view(np.uint64))


def soak(ctx, trials, seed, verbose=True):
    """`trials` random panels through every variant on context `ctx`; returns (checks, failures).
    tests/test_gpu_soak.py runs a 60-panel slice of this in the -m gpu suite."""
    import oracle_lib as ol
    from garlic_amd import abi

    rng = np.random.default_rng(seed)
    fails, checks, t0 = 0, 0, time.time()
    strip_env = os.environ.get("GARLIC_WLOD_STRIP_GROUPS")
    for trial in range(trials):
        nchr = int(rng.integers(1, 4))
        W = int(rng.choice([2, 5, 10, 15, 16, 17, 31, 32, 33, 64, 100, 129, 130]))
        sizes = [int(rng.choice([1, W - 1, W, W + 1, int(rng.integers(2 * W, 40 * W + 300))])) for _ in range(nchr)]
        sizes = [max(1, n) for n in sizes]
        nind = int(rng.choice([1, 63, 64, 65, int(rng.integers(2, 260))]))
        mg = int(rng.choice([3000, 50000, 200000]))
        chroms = [ol.random_panel(rng, n, nind, max_gap=mg, gaps=int(rng.integers(0, 5)) if n > 50 else 0,
                                  miss=float(rng.choice([0.0, 0.03, 0.3]))) for n in sizes]
        gpos = [np.cumsum(np.diff(c[2], prepend=0) * 1e-6 * rng.uniform(0.8, 1.2, size=c[2].shape[0])) for c in chroms]
        err = float(rng.choice([1e-6, 0.001, 0.05]))
        tag = (trial, sizes, W, nind, mg)
        with abi.Panel(ctx, sizes, nind) as panel:
            panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms],
                          gpos=np.concatenate(gpos))
            panel.set_freq(np.concatenate([c[1] for c in chroms]))
            panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
            lod = [ol.oracle_calc_lod(g, f, p, cs, ce, W, err, mg) for g, f, p, cs, ce in chroms]
            # full scores, aligned and dense layouts
            for pa in (32, 1):
                out = panel.lod_windows(W, err, mg, pitch_align=pa)
                for c in range(nchr):
                    checks += 1
                    if not ol.bits_equal(np.ascontiguousarray(out[c]), lod[c]):
                        fails += 1
                        print("FAIL lod", pa, c, tag)
            # thinned feed
            step = int(rng.choice([1, 3, 4, W, 2 * W + 1, 32, 64]))
            feed, per_chr = panel.lod_feed(W, err, mg, step)
            want = [ol.oracle_flatten(x, step) for x in lod]
            checks += 1
            if [len(w) for w in want] != list(per_chr) or not ol.bits_equal(feed, np.concatenate(want) if want else feed):
                fails += 1
                print("FAIL feed", step, tag)
            # coverage counts without the score matrix (garlic_roh_coverage_fused) against the oracle's inWin[] of the
            # oracle's scores; dense and 16-byte-aligned rows
            cut = float(rng.choice([-3.0, 0.0, 1.5]))
            pa = int(rng.choice([1, 8]))
            cov = panel.roh_coverage_fused(W, err, mg, cut, pitch_align=pa)
            for c in range(nchr):
                checks += 1
                if not np.array_equal(cov[c][:, :sizes[c]], ol.oracle_roh_coverage(np.ascontiguousarray(lod[c]), W, cut)):
                    fails += 1
                    print("FAIL fused coverage", cut, pa, c, tag)
            # ... and past the counts: the ROH segments (garlic_roh_segments) against the oracle's walk over the oracle's counts
            frac = float(rng.choice([1e-9, 0.25, 0.5, 1.0]))
            segs = [tuple(int(v) for v in r) for r in panel.roh_segments(W, err, mg, cut, frac)]
            want_segs = []
            for c, (g, f, p, cs, ce) in enumerate(chroms):
                covc = ol.oracle_roh_coverage(np.ascontiguousarray(lod[c]), W, cut)
                want_segs += [(i, c, a, b) for i, a, b in ol.oracle_roh_segments(covc, p, cs, ce, W, mg, frac)]
            checks += 1
            if segs != sorted(want_segs):
                fails += 1
                print("FAIL roh segments", cut, frac, len(segs), len(want_segs), tag)
            # the subset feed (--kde-subsample): drawn individuals in drawn order
            if nind > 1:
                idx = rng.choice(nind, size=int(rng.integers(1, min(nind, 40) + 1)), replace=False).astype(np.int32)
                feed, per_chr = panel.lod_feed(W, err, mg, step, ind_idx=idx)
                want = [ol.oracle_flatten_subset(x, step, idx) for x in lod]
                checks += 1
                if [len(w) for w in want] != list(per_chr) or not ol.bits_equal(feed, np.concatenate(want) if want else feed):
                    fails += 1
                    print("FAIL subset feed", step, tag)
            # LD weights, unphased and phased, with a subsample; wLOD from them
            if W <= 130 and sum(sizes) * W <= 250000:
                sub = None if rng.integers(0, 2) else np.sort(rng.choice(nind, size=int(rng.integers(1, nind + 1)), replace=False)).astype(np.int32)
                ld = panel.compute_ld(W, sub_idx=sub)
                want_ld = np.concatenate([ol.oracle_hr2_ld(c[0], W, idx=sub) for c in chroms], axis=0)
                checks += 1
                if not same(ld, want_ld):
                    fails += 1
                    print("FAIL hr2", tag)
                phase = rng.integers(0, 2, size=(sum(sizes), nind)).astype(np.uint8)
                panel.set_phase(phase)
                r2 = panel.compute_ld(W, sub_idx=sub, phased=True)
                o, parts = 0, []
                for c in chroms:
                    parts.append(ol.oracle_r2_ld(c[0], phase[o:o + c[0].shape[0]], c[1], W, idx=sub))
                    o += c[0].shape[0]
                checks += 1
                if not same(r2, np.concatenate(parts, axis=0)):
                    fails += 1
                    print("FAIL r2", tag)
            # wLOD (synthetic weights) and TGLS
            lds = [rng.uniform(1.0, max(2.0, W / 4.0), size=(n, W)) for n in sizes]
            panel.set_ld(W, np.concatenate(lds, axis=0))
            out = panel.wlod_windows(W, err, mg, 7, 1e-9, pitch_align=32)
            for c, (g, f, p, cs, ce) in enumerate(chroms):
                checks += 1
                if not ol.bits_equal(np.ascontiguousarray(out[c]), ol.oracle_calc_wlod(g, f, p, gpos[c], lds[c], cs, ce, W, err, mg, 1e-9, 7)):
                    fails += 1
                    print("FAIL wlod", c, tag)
            # ... and the weighted coverage counts without the score matrix
            cutw = float(rng.choice([-2.0, 0.0, 1.0]))
            covw = panel.roh_coverage_fused(W, err, mg, cutw, pitch_align=int(rng.choice([1, 8])), weighted=True)
            for c, (g, f, p, cs, ce) in enumerate(chroms):
                checks += 1
                wantw = ol.oracle_calc_wlod(g, f, p, gpos[c], lds[c], cs, ce, W, err, mg, 1e-9, 7)
                if not np.array_equal(covw[c][:, :sizes[c]], ol.oracle_roh_coverage(np.ascontiguousarray(wantw), W, cutw)):
                    fails += 1
                    print("FAIL weighted fused coverage", cutw, c, tag)
            fracw = float(rng.choice([0.25, 1.0]))
            segsw = [tuple(int(v) for v in r) for r in panel.roh_segments(W, err, mg, cutw, fracw, weighted=True)]
            want_segs = []
            for c, (g, f, p, cs, ce) in enumerate(chroms):
                wantw = ol.oracle_calc_wlod(g, f, p, gpos[c], lds[c], cs, ce, W, err, mg, 1e-9, 7)
                covc = ol.oracle_roh_coverage(np.ascontiguousarray(wantw), W, cutw)
                want_segs += [(i, c, a, b) for i, a, b in ol.oracle_roh_segments(covc, p, cs, ce, W, mg, fracw)]
            checks += 1
            if segsw != sorted(want_segs):
                fails += 1
                print("FAIL weighted roh segments", cutw, fracw, len(segsw), len(want_segs), tag)
            if rng.integers(0, 2):       # a dictionary of likelihood values ...
                gl = [rng.choice([1e-16, 1e-3, 0.01, 0.2, 1.0], size=c[0].shape) for c in chroms]
            else:                        # ... or any doubles (continuous mode: lod() on the device)
                gl = [np.where(rng.random(c[0].shape) < 0.02, rng.choice([0.0, 1.0, 1e-300, 0.5], size=c[0].shape),
                               10.0 ** rng.uniform(-9, 0, size=c[0].shape)) for c in chroms]
            # many short strips for the strip form of the GL-weighted kernel, now and then
            if rng.integers(0, 2):
                os.environ["GARLIC_WLOD_STRIP_GROUPS"] = str(int(rng.integers(1, 12)))
            else:
                os.environ.pop("GARLIC_WLOD_STRIP_GROUPS", None)
            panel.set_gl(np.concatenate(gl, axis=0))
            out = panel.lod_windows(W, err, mg, use_gl=True, pitch_align=32)
            for c, (g, f, p, cs, ce) in enumerate(chroms):
                checks += 1
                if not ol.bits_equal(np.ascontiguousarray(out[c]), ol.oracle_calc_lod(g, f, p, cs, ce, W, err, mg, gl=gl[c])):
                    fails += 1
                    print("FAIL tgls", c, tag)
            # coverage counts from the TGLS chain's bits (dictionary mode: the ring kernel; continuous: whichever path)
            cutg = float(rng.choice([-2.0, 0.5]))
            covg = panel.roh_coverage_fused(W, err, mg, cutg, pitch_align=8, use_gl=True)
            for c, (g, f, p, cs, ce) in enumerate(chroms):
                checks += 1
                wantg = ol.oracle_calc_lod(g, f, p, cs, ce, W, err, mg, gl=gl[c])
                if not np.array_equal(covg[c][:, :sizes[c]], ol.oracle_roh_coverage(np.ascontiguousarray(wantg), W, cutg)):
                    fails += 1
                    print("FAIL tgls fused coverage", cutg, c, tag)
            # ... and the ROH segments from the same bits
            fracg = float(rng.choice([0.25, 0.7]))
            segsg = [tuple(int(v) for v in r) for r in panel.roh_segments(W, err, mg, cutg, fracg, use_gl=True)]
            want_segs = []
            for c, (g, f, p, cs, ce) in enumerate(chroms):
                wantg = ol.oracle_calc_lod(g, f, p, cs, ce, W, err, mg, gl=gl[c])
                covc = ol.oracle_roh_coverage(np.ascontiguousarray(wantg), W, cutg)
                want_segs += [(i, c, a, b) for i, a, b in ol.oracle_roh_segments(covc, p, cs, ce, W, mg, fracg)]
            checks += 1
            if segsg != sorted(want_segs):
                fails += 1
                print("FAIL tgls roh segments", cutg, fracg, len(segsg), len(want_segs), tag)
            out = panel.wlod_windows(W, err, mg, 7, 1e-9, pitch_align=32, use_gl=True)
            for c, (g, f, p, cs, ce) in enumerate(chroms):
                checks += 1
                if not ol.bits_equal(np.ascontiguousarray(out[c]), ol.oracle_calc_wlod(g, f, p, gpos[c], lds[c], cs, ce, W, err, mg, 1e-9, 7, gl=gl[c])):
                    fails += 1
                    print("FAIL wlod+gl", c, tag)
            # a multi-size feed: the sizes the single calls above have checked
            W2 = int(rng.choice([4, 9, 40, 70]))
            feeds, per_chr = panel.lod_feed_multi([W, W2], err, mg, steps=[max(4, W), 5])
            for k, (Wk, sk) in enumerate(((W, max(4, W)), (W2, 5))):
                want = [ol.oracle_flatten(ol.oracle_calc_lod(g, f, p, cs, ce, Wk, err, mg), sk) for g, f, p, cs, ce in chroms]
                checks += 1
                if [len(w) for w in want] != list(per_chr[k]) or not ol.bits_equal(feeds[k], np.concatenate(want)):
                    fails += 1
                    print("FAIL feed_multi", Wk, sk, tag)
            # liveness book-keeping: no strip launch had to be repaired, no count item timed out
            st = panel.stats()
            checks += 1
            if st["n_stall_reruns"] or st["n_count_timeouts"]:
                fails += 1
                print("FAIL liveness counters", st["n_stall_reruns"], st["n_count_timeouts"], tag)
        if verbose and trial % 10 == 9:
            print(f"trial {trial + 1}: {checks} checks, {fails} failures, {time.time() - t0:.0f} s", flush=True)
    if strip_env is None:
        os.environ.pop("GARLIC_WLOD_STRIP_GROUPS", None)
    else:
        os.environ["GARLIC_WLOD_STRIP_GROUPS"] = strip_env
    return checks, fails


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from garlic_amd import abi
    checks, fails = soak(abi.Context(0), args.trials, args.seed)
    print(f"soak: {args.trials} panels, {checks} checks, {fails} failures")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
