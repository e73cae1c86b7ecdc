#!/usr/bin/env python3
"""End-to-end rate of the C ABI with HOST buffers (what GARLIC's drop-in call pays): genotypes
uploaded from host int16 rows, scores copied back into a host array.  C2-shaped panel by default.
    python tools/bench_pcie.py [--snps 1000000] [--inds 1000] [--winsize 100]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snps", type=int, default=1_000_000)
    ap.add_argument("--inds", type=int, default=1000)
    ap.add_argument("--winsize", type=int, default=100)
    args = ap.parse_args()
    import torch
    from garlic_amd import abi, synth

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    spec = synth.PanelSpec(args.snps, seed=20260102, max_gap=200000)
    geno = np.empty((args.snps, args.inds), dtype=np.int16)
    for l0, g in synth.genotype_chunks(spec, args.inds, dev):
        geno[l0:l0 + g.shape[0]] = g.cpu().numpy()
    ctx = abi.Context(0)
    t0 = time.perf_counter()
    panel = abi.Panel(ctx, spec.chr_nloci, args.inds)
    panel.set_map(spec.pos, spec.centro_start, spec.centro_end)
    panel.set_freq(spec.freq)
    panel.set_genotypes(geno)
    t_up = time.perf_counter() - t0
    base, pitch, total = panel.out_layout(1, args.inds)
    out = np.empty(total, dtype=np.float64)
    out[::512] = 0  # touch the pages once: the timed call should not pay first-touch faults
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        abi.check(abi.lib().garlic_lod_windows(panel.handle, args.winsize, 0.001, 200000, 0, 0, args.inds, 1,
                                               abi._vp(out.ctypes.data), abi.HOST))
        times.append(time.perf_counter() - t0)
    st = panel.stats()
    # the explore / auto-winsize form: only the thinned KDE feed comes back
    panel.lod_feed(args.winsize, 0.001, 200000, args.winsize)
    t0 = time.perf_counter()
    feed, _ = panel.lod_feed(args.winsize, 0.001, 200000, args.winsize)
    t_feed = time.perf_counter() - t0
    win = args.snps * args.inds
    print(json.dumps({"snps": args.snps, "inds": args.inds, "winsize": args.winsize,
                      "upload_s (int16 genotypes, pack on device)": t_up,
                      "lod_windows_host_output_s": min(times),
                      "of_which_device_ms": st["total_ms"], "chain_kernel_ms": st["chain_kernel_ms"],
                      "d2h_GBps": win * 8 / 1e9 / max(1e-9, min(times) - st["chain_kernel_ms"] * 1e-3),
                      "lod_windows_per_s_pcie_inclusive": win / args.winsize / min(times),
                      "lod_feed_call_s (scores + thinning on device, feed of %d values back)" % feed.shape[0]: t_feed,
                      "lod_windows_per_s_feed_only": win / args.winsize / t_feed}))


if __name__ == "__main__":
    main()
