#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench line): the TGLS (--gl-type GQ) and wLOD
(--weighted) variants of the path on a synthetic panel, one JSON line each.

    python tools/bench_variants.py [--snps 200000] [--inds 1000] [--winsize 100] [--steps 5]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snps", type=int, default=200000)
    ap.add_argument("--inds", type=int, default=1000)
    ap.add_argument("--winsize", type=int, default=100)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--modes", default="lod,tgls,wlod")
    args = ap.parse_args()

    import torch
    from garlic_amd import abi, synth

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    nloci, nind, W = args.snps, args.inds, args.winsize
    error, max_gap = 0.001, 200000
    spec = synth.PanelSpec(nloci, seed=20260105, max_gap=max_gap)
    ctx = abi.Context(0)
    panel = abi.Panel(ctx, spec.chr_nloci, nind)
    panel.set_map(spec.pos, spec.centro_start, spec.centro_end, gpos=spec.gpos)
    panel.set_freq(spec.freq)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    for l0, g in synth.genotype_chunks(spec, nind, dev):
        torch.cuda.synchronize()
        panel.set_genotypes_device(g.data_ptr(), g.shape[1], l0, g.shape[0])
        if "tgls" in args.modes or "wlodgl" in args.modes:
            # GQ ~ integer U{3..60}; error = 10^max(-10, -GQ/10)   (garlic-data.cpp:1557)
            gq = torch.randint(3, 61, g.shape, generator=gen, device=dev).to(torch.float64)
            gl = torch.pow(torch.tensor(10.0, dtype=torch.float64, device=dev), -gq / 10.0)
            torch.cuda.synchronize()
            panel.set_gl_device(gl.data_ptr(), gl.shape[1], l0, gl.shape[0])
            del gq, gl
    del g
    base, pitch, total = panel.out_layout(32, nind)
    out = torch.empty(total, dtype=torch.float64, device=dev)
    if "wlod" in args.modes:
        # synthetic LD weights U(1, W/4) until the LD kernel exists
        ld = 1.0 + (W / 4.0 - 1.0) * torch.rand((nloci, W), generator=gen, device=dev, dtype=torch.float64)
        torch.cuda.synchronize()
        panel.set_ld_device(W, ld.data_ptr())

    def run(mode):
        if mode == "lod":
            panel.lod_windows_device(out.data_ptr(), W, error, max_gap)
        elif mode == "tgls":
            panel.lod_windows_device(out.data_ptr(), W, error, max_gap, use_gl=True)
        elif mode == "wlodgl":
            panel.wlod_windows_device(out.data_ptr(), W, error, max_gap, 7, 1e-9, use_gl=True)
        else:
            panel.wlod_windows_device(out.data_ptr(), W, error, max_gap, 7, 1e-9)

    if "ld" in args.modes.split(","):
        # LD weights (calcHR2LD) from the resident genotypes, all individuals; wall clock of the call
        import time
        panel.compute_ld(W, want_output=False)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            panel.compute_ld(W, want_output=False)
        dt = (time.perf_counter() - t0) / args.steps
        print(json.dumps({"mode": "ld", "snps": nloci, "inds": nind, "winsize": W, "call_ms": dt * 1e3,
                          "loci_per_s": nloci / dt, "pair_counts_per_s": nloci * (W - 1) / dt,
                          "ld_sums_terms_per_s": nloci * W * W / dt}))
    if "feed" in args.modes.split(","):
        # KDE feed (garlic_lod_feed, unweighted, step = W): chain kernel with the thinned write-out, then
        # the compaction.  chain_kernel_ms = the chain kernel alone; call_ms = wall clock incl. the D2H
        # of the feed.
        import time
        panel.lod_feed(W, error, max_gap, W, copy=False)
        ms, wall = [], []
        for _ in range(args.steps):
            t0 = time.perf_counter()
            feed, _ = panel.lod_feed(W, error, max_gap, W, copy=False)
            wall.append(time.perf_counter() - t0)
            ms.append(panel.stats()["chain_kernel_ms"])
        k = float(np.mean(ms))
        print(json.dumps({"mode": "feed", "snps": nloci, "inds": nind, "winsize": W, "step": W,
                          "feed_values": int(feed.shape[0]), "chain_kernel_ms": k,
                          "call_ms": float(np.mean(wall)) * 1e3,
                          "sliding_windows_per_s": nloci * nind / (k * 1e-3),
                          }))
    for mode in [m for m in args.modes.split(",") if m not in ("ld", "feed")]:
        run(mode)
        ms = []
        for _ in range(args.steps):
            run(mode)
            ms.append(panel.stats()["chain_kernel_ms"])
        k = float(np.mean(ms))
        win = nloci * nind
        line = {"mode": mode, "snps": nloci, "inds": nind, "winsize": W, "kernel_ms": k,
                "sliding_windows_per_s": win / (k * 1e-3),
                "lod_windows_per_s": win / W / (k * 1e-3),
                "out_GBps": win * 8 / (k * 1e-3) / 1e9}
        # what bounds the kernel: HBM bytes per sliding window for the chains,
        # separately rounded FP64 multiply + add pairs for the weighted sums
        if mode in ("lod", "tgls"):
            per_win = 8.25 if mode == "lod" else 16.25
            a = win * per_win / (k * 1e-3) / 1e9
            line["roofline"] = {"bound": "hbm", "achieved": a, "peak": 8000.0, "unit": "GB/s", "frac": a / 8000.0,
                                "algorithmic_bytes_per_window": per_win}
        else:
            pairs = win * W / (k * 1e-3)
            line["roofline"] = {"bound": "fp64 valu (v_mul_f64 + v_add_f64 per term, no FMA)",
                                "achieved": 2 * pairs / 1e12, "peak": 78.6 / 2, "unit": "TFLOP/s",
                                "frac": 2 * pairs / 1e12 / (78.6 / 2),
                                "measured_ceiling_TFLOPs": 35.8, "frac_of_measured_ceiling": 2 * pairs / 1e12 / 35.8}
        print(json.dumps(line))


if __name__ == "__main__":
    main()
