#!/usr/bin/env python3
"""bench.py -- Phase-I window LOD throughput on MI355X (BASELINE.json metric).

A "step" is one full pass of the hot path (calcLODWindows, reference src/garlic-roh.cpp:279) over a
synthetic SNP x individual panel that is already resident in HBM as 2-bit packed genotypes: the MISSING
fill and the LOD chain kernel, writing every window score of every individual (FP64, individual-major)
into HBM.  (The work plan -- gap / centromere segments, runs, the longest-first work list -- is made by the
first call and stays resident; the timed passes, which repeat its arguments, only enqueue the two kernels.)

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|ns|c3w100|small] [--mode lod|wlod|tgls]
                    [--also auto|none|leg,leg..] [--no-cpu]

N = 1 (default): the workload BASELINE.json's metric is quoted on, configs[1] (C2: 1M SNPs x 1000
individuals, --winsize 100, unweighted).  N > 1 is launched by the driver through torch.distributed.run, one
rank per GPU, and runs the north star's panel: 10M SNPs, 1250 individuals per rank (N = 8: the configured
10M x 10k panel; --workload ns runs one such shard at N = 1).  Individuals shard across ranks, every rank
owns all SNPs of its individuals (weak scaling); there is no collective on the data path -- only the timing
barrier, and with --mode wlod one integer all-reduce of the LD pair counts before the timed region.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel against its bound with the kernel's
HIP-event time measured live on the stream it runs on; `cpu_baseline` times the reference's own calcLOD
(oracle/_ref, built from the reference sources in the build container) or, if that library is absent, the C
port in oracle/, on a bounded sample of the same panel on the host cores -- and the GPU scores of those
individuals are compared with it bit for bit.  At N = 1 the line also carries, under `also`, the other
configured workloads with their own roofline objects (C3: 5M x 5k x four window sizes, full scores and
thinned feed; C4 / C5 per-GPU shard, 10M x 1250: unweighted, LD weights, wLOD, TGLS, GL-weighted wLOD) and
`end_to_end` (host int16 genotypes in -> host doubles out through the C ABI, PCIe included); legs are skipped,
and say so, once the run has used its time budget (--also-budget-s).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (nloci, inds per GPU, winsize, seed offset, BASELINE.json config it is)
    "c2": (1_000_000, 1000, 100, 1, "synthetic 1M SNPs x 1k inds, --winsize 100 --overlap-frac 0.25, unweighted LOD"),
    "ns": (10_000_000, 1250, 100, 3, "synthetic 10M SNPs x 10k inds sharded over 8 GPUs: 1250 inds per GPU (configs 4/5 panel)"),
    "c3w100": (5_000_000, 5000, 100, 2, "synthetic 5M SNPs x 5k inds, one window size (100) of config 3"),
    "small": (100_000, 256, 100, 0, "smoke-sized panel (not a BASELINE config)"),
}
BYTES_LOD = 8.25     # 0.25 B 2-bit genotype in + 8 B double out per sliding window
BYTES_TGLS = 16.25   # 8 B term + 0.25 B genotype + 8 B score ("GL as doubles" row)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TFLOPS = 39.3      # FP64 vector peak counted as separate mul + add (78.6 TFLOP/s counts FMAs; wLOD may not fuse)
ERROR, MAX_GAP, M_GEN, MU = 0.001, 200000, 7, 1e-9
T_START = time.perf_counter()


def hbm_roofline(kernel, bytes_per_launch, kernel_ms, **extra):
    a = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "kernel": kernel, "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
         "traffic": None, "traffic_source": "not measured in this run: PMC passes are separate rocprofv3 runs, "
                                            "see profiles/ (bytes per launch of the same kernel and panel)",
         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": bytes_per_launch}
    r.update(extra)
    return r


class ClockSampler:
    """Shader clock and socket power from rocm-smi while a kernel loops (a thread polling every ~0.3 s).  The FP64-bound
    wLOD kernels run power-capped: `peak` in their roofline object is the spec figure at 2.4 GHz, this says what clock
    the chip actually held ("wLOD is power-bound")."""

    def __init__(self, smi_index=0):
        import threading
        self.idx, self.samples, self._stop = smi_index, [], threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re
        import subprocess
        while not self._stop.is_set():
            try:
                txt = subprocess.run(["rocm-smi", "-d", str(self.idx), "--showclocks", "--showpower"], capture_output=True,
                                     text=True, timeout=5).stdout
                m = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", txt)
                w = re.search(r"Power \(W\):\s*([0-9.]+)", txt)
                if m:
                    self.samples.append((float(m.group(1)), float(w.group(1)) if w else float("nan")))
            except Exception:
                return
            self._stop.wait(0.2)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join(timeout=10)

    def summary(self):
        busy = [x for x in self.samples[1:] if x[0] > 1000.0]         # first sample: the clock is still ramping
        if len(busy) < 2:
            return None
        return {"sclk_mhz_under_load": float(np.mean([x[0] for x in busy])), "socket_power_w": float(np.mean([x[1] for x in busy])),
                "samples": len(busy), "source": "rocm-smi -d %d --showclocks --showpower, polled while the kernel looped" % self.idx}


def clock_under_load(call, sync, seconds=2.5):
    """loops `call` for a few seconds with the sampler running; None when rocm-smi is not there"""
    import shutil
    if not shutil.which("rocm-smi") or os.environ.get("GARLIC_BENCH_NO_CLOCK"):      # (no child processes under rocprofv3 --pmc)
        return None
    with ClockSampler() as cs:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(4):
                call()
            sync()
    return cs.summary()


def fp64_roofline(kernel, windows, W, kernel_ms, **extra):
    flops = 2.0 * windows * W        # one v_mul_f64 + one v_add_f64 per (window, term): the product rounds before the add
    a = flops / (kernel_ms * 1e-3) / 1e12
    r = {"bound": "fp64 valu (separately rounded multiply + add per term: no FMA, no MFMA)", "kernel": kernel,
         "achieved": a, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": a / FP64_PEAK_TFLOPS, "traffic": None,
         "kernel_ms": kernel_ms, "algorithmic_flops_per_launch": flops}
    clk = extra.pop("clock", None)
    if clk:
        r["clock_under_load"] = clk
        r["frac_at_measured_clock"] = r["frac"] * 2400.0 / clk["sclk_mhz_under_load"]
    r.update(extra)
    return r


def cpu_baseline(spec, geno_sample, W, gpu_rows, seconds_budget=25.0, all_cores=True):
    """Times the CPU path on `geno_sample` (int16 [nloci][n_s]) one chromosome at a time and checks the GPU
    rows against it.  Returns the cpu_baseline JSON object.  (At N > 1 every rank runs it on a few of its own
    individuals with a small budget: the bit check per rank; rank 0's timing is the line's cpu_baseline.)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol

    use_ref = ol.have_ref()
    n_s = geno_sample.shape[1]
    t_total, windows, mismatches = 0.0, 0, 0
    for c in range(spec.nchr):
        lo, hi = int(spec.chr_off[c]), int(spec.chr_off[c + 1])
        g = np.ascontiguousarray(geno_sample[lo:hi])
        args = (g, spec.freq[lo:hi], spec.pos[lo:hi], int(spec.centro_start[c]), int(spec.centro_end[c]), W, ERROR, MAX_GAP)
        t0 = time.perf_counter()
        want = ol.ref_calc_lod(*args) if use_ref else ol.oracle_calc_lod(*args)
        t_total += time.perf_counter() - t0
        windows += (hi - lo) * n_s
        mismatches += ol.count_mismatch(np.ascontiguousarray(gpu_rows[c]), want)
        if t_total > seconds_budget:
            break
    # (ii) all host cores, SURVEY 8(d): independent slices of individuals, one per core, through the C port
    # (oracle/) -- the reference itself has no threaded calcLOD.  One chromosome's worth of the same sample.
    # The GPU box gives one GPU's job a share of 16 host cores whatever the affinity mask says.
    single = {
        "value": windows / W / t_total, "unit": "LOD-windows/s", "cores": 1,
        "kind": "reference" if use_ref else "port",
        "sample": f"{n_s} individuals x {windows // n_s} SNPs ({c + 1} of {spec.nchr} chromosomes) of the same panel, "
                  f"single thread (calcLOD is single-threaded in the reference), {t_total:.1f} s",
        "sliding_windows_per_s": windows / t_total,
        "gpu_bit_mismatches_on_sample": int(mismatches),
        "values_compared": int(windows),
    }
    if not all_cores:
        return single
    ncores = min(len(os.sched_getaffinity(0)), int(os.environ.get("GARLIC_BENCH_CORES", "16")))
    lo, hi = int(spec.chr_off[0]), int(spec.chr_off[1])
    a0 = (np.ascontiguousarray(geno_sample[lo:hi]), spec.freq[lo:hi], spec.pos[lo:hi], int(spec.centro_start[0]),
          int(spec.centro_end[0]), W, ERROR, MAX_GAP)
    ol.oracle_calc_lod(*a0, threads=ncores)                 # page in, spin up the team
    reps, t_all = 0, 0.0
    while t_all < 3.0 and reps < 20:
        t0 = time.perf_counter()
        ol.oracle_calc_lod(*a0, threads=ncores)
        t_all += time.perf_counter() - t0
        reps += 1
    return dict(single, all_cores={"value": (hi - lo) * n_s * reps / W / t_all, "unit": "LOD-windows/s", "cores": ncores, "kind": "port",
                      "sample": f"{n_s} individuals x {hi - lo} SNPs (chromosome 1), {n_s // max(1, ncores)} "
                                f"individuals per core, {reps} repetitions, {t_all:.1f} s"})


def load_panel(ctx, spec, nind, dev, ind_offset=0, n_cpu=0, gq=False):
    """device-resident panel of `nind` individuals drawn on the device; n_cpu: also keep the first n_cpu
    individuals' genotypes on the host (CPU baseline); gq: GQ ~ U{3..60} likelihoods (--gl-type GQ, config 5)"""
    import torch
    from garlic_amd import abi, synth
    panel = abi.Panel(ctx, spec.chr_nloci, nind)
    panel.set_map(spec.pos, spec.centro_start, spec.centro_end, gpos=spec.gpos)
    panel.set_freq(spec.freq)
    geno_sample = np.empty((spec.nloci, n_cpu), dtype=np.int16) if n_cpu else None
    gen = torch.Generator(device=dev)
    gen.manual_seed(spec.seed * 7 + ind_offset)
    for l0, g in synth.genotype_chunks(spec, nind, dev, ind_offset=ind_offset):
        torch.cuda.synchronize()
        panel.set_genotypes_device(g.data_ptr(), g.shape[1], l0, g.shape[0])
        if n_cpu:
            geno_sample[l0:l0 + g.shape[0]] = g[:, :n_cpu].cpu().numpy()
        if gq:   # error = 10^max(-10, -GQ/10)   (garlic-data.cpp:1557)
            q = torch.randint(3, 61, g.shape, generator=gen, device=dev).to(torch.float64)
            gl = torch.pow(torch.tensor(10.0, dtype=torch.float64, device=dev), -q / 10.0)
            torch.cuda.synchronize()
            panel.set_gl_device(gl.data_ptr(), gl.shape[1], l0, gl.shape[0])
            del q, gl
    return panel, geno_sample


def timed_passes(ctx, call, steps, warmup, sync):
    """`warmup` untimed then `steps` timed passes of call(), enqueued back to back; returns (seconds for
    the timed passes, mean HIP-event ms of the dominant kernel over them)"""
    for _ in range(warmup):
        call()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        call()
    sync()
    dt = time.perf_counter() - t0
    k = ctx.recent_kernel_ms(min(steps, 32))
    return dt, float(np.mean(k))


# ------------------------------------------------------------------------------------------------ also legs
def leg_c3(ctx, dev, steps):
    """config 3: 5M SNPs x 5k individuals, --winsize-multi 50 100 200 300 on ONE resident panel: the four
    full-score passes (200 GB of scores each, into the same buffer) and the four thinned feeds
    (--auto-winsize / exploreWinsizes keep only convertWinData2DoubleData(.., step = winsize))"""
    import torch
    from garlic_amd import synth
    nloci, nind = 5_000_000, 5000
    sizes = [50, 100, 200, 300]
    spec = synth.PanelSpec(nloci, seed=20260101 + 2, max_gap=MAX_GAP)
    panel, _ = load_panel(ctx, spec, nind, dev)
    base, pitch, total = panel.out_layout(32, nind)
    out = ctx.alloc_scores(total)
    torch.cuda.synchronize()
    res = {"workload": "synthetic 5M SNPs x 5k inds, --winsize-multi 50 100 200 300 (config 3), one resident panel",
           "snps": nloci, "inds": nind, "winsizes": sizes}
    full, feed = {}, {}
    for W in sizes:
        call = lambda: panel.lod_windows_device(out.data_ptr(), W, ERROR, MAX_GAP)
        dt, k = timed_passes(ctx, call, steps, 1, torch.cuda.synchronize)
        full[W] = {"ms_per_pass": dt / steps * 1e3, "kernel_ms": k}
    k_all = sum(v["kernel_ms"] for v in full.values())
    t_all = sum(v["ms_per_pass"] for v in full.values())
    res["full_scores"] = {
        "per_winsize": full, "four_sizes_ms": t_all,
        "lod_windows_per_s": sum(nloci * nind / W for W in sizes) / (t_all * 1e-3),
        "sliding_windows_per_s": 4 * nloci * nind / (t_all * 1e-3),
        "roofline": hbm_roofline("lod_chain_kernel", 4 * BYTES_LOD * nloci * nind, k_all,
                                 note="the four launches of one --winsize-multi call together")}
    out.free()
    torch.cuda.empty_cache()
    ctx.trim()
    for W in sizes:
        panel.lod_feed(W, ERROR, MAX_GAP, W, copy=False)             # plan + scratch
        ks, ws = [], []
        for _ in range(max(2, steps // 2)):
            t0 = time.perf_counter()
            f, _ = panel.lod_feed(W, ERROR, MAX_GAP, W, copy=False)
            ws.append(time.perf_counter() - t0)
            ks.append(panel.stats()["chain_kernel_ms"])
        feed[W] = {"kernel_ms": float(np.mean(ks)), "call_ms": float(np.mean(ws)) * 1e3, "feed_values": int(f.shape[0])}
    k_all = sum(v["kernel_ms"] for v in feed.values())
    bytes_feed = sum((0.25 + 8.0 / W) * nloci * nind for W in sizes)
    # the whole --winsize-multi flow in one call: every size on a stream of its own, feeds fetched in order
    panel.lod_feed_multi(sizes, ERROR, MAX_GAP, copy=False)      # scratch, streams
    ws = []
    for _ in range(max(2, steps // 2)):
        t0 = time.perf_counter()
        fm, _ = panel.lod_feed_multi(sizes, ERROR, MAX_GAP, copy=False)
        ws.append(time.perf_counter() - t0)
    feed_bytes = 8.0 * sum(int(f.shape[0]) for f in fm)
    res["thinned_feed"] = {
        "per_winsize": feed, "four_sizes_kernel_ms": k_all,
        "four_sizes_call_ms": float(np.mean(ws)) * 1e3,
        "four_sizes_call_note": "one garlic_lod_feed_multi call: the sizes' kernels overlap with each other's tails and with "
                                "the feeds' way over PCIe (%.2f GB to host memory)" % (feed_bytes / 1e9),
        "four_single_calls_ms": sum(v["call_ms"] for v in feed.values()),
        "feed_gb": feed_bytes / 1e9,
        "sliding_windows_per_s": 4 * nloci * nind / (k_all * 1e-3),
        "roofline": hbm_roofline("lod_feed_kernel", bytes_feed, k_all,
                                 note="0.25 + 8/W B per window (SURVEY 8(d) 'thinned output'); bound by the two dependent FP64 adds and "
                                      "the two term look-ups per window and lane (vector-instruction issue), not by memory: "
                                      "fp64_add_frac = the adds alone against the FP64 add rate"),
        "fp64_add_frac": (2.0 * 4 * nloci * nind / (k_all * 1e-3)) / (FP64_PEAK_TFLOPS * 1e12)}
    panel.close()
    return res


def leg_ns(ctx, dev, steps):
    """configs 4 and 5, one GPU's shard: 10M SNPs x 1250 individuals -- unweighted scores, LD weights
    (all individuals; --ld-subsample 500), wLOD, TGLS (--gl-type GQ), GL-weighted wLOD"""
    import torch
    from garlic_amd import synth
    nloci, nind, W = 10_000_000, 1250, 100
    spec = synth.PanelSpec(nloci, seed=20260101 + 3, max_gap=MAX_GAP)
    panel, _ = load_panel(ctx, spec, nind, dev, gq=True)
    base, pitch, total = panel.out_layout(32, nind)
    out = ctx.alloc_scores(total)
    torch.cuda.synchronize()
    win = nloci * nind
    res = {"workload": "synthetic 10M SNPs x 1250 inds (one GPU's shard of the 10M x 10k panel of configs 4 and 5), --winsize 100",
           "snps": nloci, "inds": nind, "winsize": W}

    def rate(k_ms, dt):
        return {"kernel_ms": k_ms, "ms_per_pass": dt / steps * 1e3, "sliding_windows_per_s": win / (dt / steps),
                "lod_windows_per_s": win / W / (dt / steps)}

    dt, k = timed_passes(ctx, lambda: panel.lod_windows_device(out.data_ptr(), W, ERROR, MAX_GAP), steps, 1,
                         torch.cuda.synchronize)
    res["lod"] = dict(rate(k, dt), roofline=hbm_roofline("lod_chain_kernel", BYTES_LOD * win, k))
    # the first half of assembleROHWindows on the resident scores (garlic_roh_coverage): 8 B of scores in, 2 B of counts out
    _, _, tcov = panel.out_layout(8, nind)              # rows of whole 16-B pieces: the kernel stores eight counts at a time
    cov = torch.empty(tcov, dtype=torch.int16, device=dev)
    torch.cuda.synchronize()
    panel.roh_coverage_device(out.data_ptr(), W, 2.5, cov.data_ptr(), inwin_pitch_align=8)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        panel.roh_coverage_device(out.data_ptr(), W, 2.5, cov.data_ptr(), inwin_pitch_align=8)     # synchronous call: wall = kernel + two tiny uploads
        ts.append(time.perf_counter() - t0)
    tcv = float(np.min(ts))
    res["roh_coverage"] = {"call_ms": tcv * 1e3, "sliding_windows_per_s": win / tcv,
                           "roofline": hbm_roofline("roh_coverage_kernel", 10.0 * win, tcv * 1e3,
                                                    note="10 B per window (8 B score in, 2 B count out); timed as the whole synchronous "
                                                         "call (wall clock, best of 3): kernel + two small uploads")}
    # ... and the same counts WITHOUT the scores (garlic_roh_coverage_fused: the chain leaves one bit per window, a second
    # kernel counts the bits; 2 B per window leave the chip, no score matrix resident) -- what GARLIC's final pass needs when --raw-lod is not asked for
    _, _, tcov8 = panel.out_layout(8, nind)
    cov8 = torch.empty(tcov8, dtype=torch.int16, device=dev)
    torch.cuda.synchronize()
    panel.roh_coverage_fused_device(W, ERROR, MAX_GAP, 2.5, cov8.data_ptr(), pitch_align=8)
    tf = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        panel.roh_coverage_fused_device(W, ERROR, MAX_GAP, 2.5, cov8.data_ptr(), pitch_align=8)
        tf.append(time.perf_counter() - t0)
    tfu = float(np.min(tf))
    b1, p1, _ = panel.out_layout(1, nind)
    b8, p8, _ = panel.out_layout(8, nind)
    same = all(bool(torch.equal(cov[b1[c]: b1[c] + nind * p1[c]].view(nind, p1[c])[:, :spec.chr_nloci[c]],
                                cov8[b8[c]: b8[c] + nind * p8[c]].view(nind, p8[c])[:, :spec.chr_nloci[c]]))
               for c in range(len(b1)))
    res["roh_coverage_fused"] = {"call_ms": tfu * 1e3, "sliding_windows_per_s": win / tfu,
                                 "scores_then_counts_ms": res["lod"]["kernel_ms"] + tcv * 1e3,
                                 "equals_scores_then_counts": same,
                                 "roofline": hbm_roofline("lod_bits_kernel + cov_counts_from_bits_kernel", 2.5 * win, tfu * 1e3,
                                                          note="2.5 B per window (0.25 B genotype in, 1 bit out and in again, 2 B count out): not HBM-bound -- "
                                                               "the first kernel is the hand-scheduled chain of the thinned feed with a compare and an "
                                                               "add-with-carry per window (10.1 instructions per window on the longest run's wave), the "
                                                               "second a plain pass over the bits; no 8 B per window of scores written, read or resident; "
                                                               "timed as the whole call (wall clock incl. its scratch allocations)")}
    # ... and past the counts: the ROH segments themselves (garlic_roh_segments: the bits become "SNP is covered by >=
    # OVERLAP_FRAC x winsize qualifying windows" bits and a list of (individual, chromosome, first, last) -- all of
    # assembleROHWindows on the device, a few MB to the host instead of 25 GB of counts)
    from garlic_amd import abi
    seg_cap = 16_000_000
    seg_buf = np.empty((seg_cap, 4), dtype=np.int32)
    n_seg = ctypes.c_int64()
    seg_args = (panel.handle, W, ERROR, MAX_GAP, 0, 0, M_GEN, MU, 2.5, 0.25, ctypes.c_void_p(seg_buf.ctypes.data), seg_cap, ctypes.byref(n_seg))
    abi.check(abi.lib().garlic_roh_segments(*seg_args))
    tsg = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        abi.check(abi.lib().garlic_roh_segments(*seg_args))
        tsg.append(time.perf_counter() - t0)
    res["roh_segments"] = {"call_ms": float(np.min(tsg)) * 1e3, "sliding_windows_per_s": win / float(np.min(tsg)),
                           "n_segments": int(n_seg.value), "bytes_to_host": int(n_seg.value) * 16,
                           "counts_bytes_it_replaces": int(tcov8) * 2, "cutoff": 2.5, "overlap_frac": 0.25,
                           "note": "lod_bits_kernel + roh_mask_from_bits_kernel + roh_segments_from_mask_kernel + the list to the host, "
                                   "sorted (wall clock of the whole call, best of 3); against roh_coverage_fused.call_ms, whose counts "
                                   "would still have to cross PCIe and be walked per individual on the host"}
    del cov, evs, cov8, seg_buf
    torch.cuda.empty_cache()      # 25 GB of counts: the likelihood legs below need the room
    # LD weights: integer pair counts (AND + popcount on bit planes) + W^2 ordered FP64 adds per window start
    for name, sub in (("ld_all_individuals", None),
                      ("ld_subsample_500", np.sort(np.random.default_rng(1).choice(nind, 500, replace=False)).astype(np.int32))):
        panel.compute_ld(W, sub_idx=sub, want_output=False)
        ts = []
        for _ in range(max(2, steps // 2)):
            t0 = time.perf_counter()
            panel.compute_ld(W, sub_idx=sub, want_output=False)
            ts.append(time.perf_counter() - t0)
        t = float(np.mean(ts))
        k_sum = float(np.mean(ctx.recent_kernel_ms(len(ts))))       # the ordered-sum kernel of each of those calls
        nsub = nind if sub is None else 500
        adds = float(nloci) * W * W
        res[name] = {"call_ms": t * 1e3, "snps_per_s": nloci / t,
                     "roofline": {"bound": "fp64 valu adds (W^2 ordered adds per window start, ld_sum_col_kernel: one LDS read per "
                                           "up to 32 of them); the pair counts are banded Gram matrices of the subsample's bit "
                                           "planes on the matrix cores (ld_pair_mfma_kernel, fp4 operands), which goes on to the pair's two hr2 "
                                           "values (three FP64 divisions) and writes the combined table itself",
                                  "kernel": "ld_* (planes -- kept across calls --, pair counts + hr2 table, ordered sums + wLOD weights) -- the whole warm call",
                                  "achieved": adds / t / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s (adds only; one separately rounded FP64 operation = one op, as in every other FP64 leg)",
                                  "frac": adds / t / 1e12 / FP64_PEAK_TFLOPS, "traffic": None,
                                  "ordered_adds_per_call": adds,
                                  "dominant_kernel": {"kernel": "ld_sum_col_kernel (ordered sums + wLOD weights)", "kernel_ms": k_sum,
                                                      "achieved": adds / (k_sum * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS,
                                                      "unit": "TFLOP/s (adds only)",
                                                      "frac": adds / (k_sum * 1e-3) / 1e12 / FP64_PEAK_TFLOPS},
                                  "popcounts_per_call": float(nloci) * (W - 1) * 2 * ((nsub + 63) // 64)}}
    dt, k = timed_passes(ctx, lambda: panel.wlod_windows_device(out.data_ptr(), W, ERROR, MAX_GAP, M_GEN, MU), steps, 1,
                         torch.cuda.synchronize)
    clk = clock_under_load(lambda: panel.wlod_windows_device(out.data_ptr(), W, ERROR, MAX_GAP, M_GEN, MU), torch.cuda.synchronize)
    res["wlod"] = dict(rate(k, dt), roofline=fp64_roofline("wlod_tile2_kernel", win, W, k, clock=clk))
    # the weighted final pass without its score matrix: the same kernel leaves 16 bits per individual and group instead of
    # 16 scores (garlic_roh_coverage_fused, weighted), then the counts from the bits
    cov8 = torch.empty(tcov8, dtype=torch.int16, device=dev)
    torch.cuda.synchronize()
    panel.roh_coverage_fused_device(W, ERROR, MAX_GAP, 2.5, cov8.data_ptr(), pitch_align=8, weighted=True, M=M_GEN, mu=MU)
    tw = []
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        panel.roh_coverage_fused_device(W, ERROR, MAX_GAP, 2.5, cov8.data_ptr(), pitch_align=8, weighted=True, M=M_GEN, mu=MU)
        tw.append(time.perf_counter() - t0)
    res["wlod_coverage_fused"] = {"call_ms": float(np.min(tw)) * 1e3, "scores_then_counts_ms": k + tcv * 1e3,
                                  "note": "wlod_tile2_kernel writing coverage bits (2 B per individual and 16 windows) + "
                                          "cov_counts_from_bits_kernel, against the score pass + roh_coverage_kernel"}
    del cov8
    torch.cuda.empty_cache()
    panel.release_scratch()
    dt, k = timed_passes(ctx, lambda: panel.lod_windows_device(out.data_ptr(), W, ERROR, MAX_GAP, use_gl=True), steps, 1,
                         torch.cuda.synchronize)
    res["tgls"] = dict(rate(k, dt), roofline=hbm_roofline("lod_chain_ring_kernel", BYTES_TGLS * win, k,
                                                          note="term matrix built once per panel (gl_terms_kernel), not in the pass"))
    dt, k = timed_passes(ctx, lambda: panel.wlod_windows_device(out.data_ptr(), W, ERROR, MAX_GAP, M_GEN, MU, use_gl=True),
                         steps, 1, torch.cuda.synchronize)
    # TGLS final pass without its score matrix: the ring chain leaves a dword of coverage bits per lane and tile
    # (the counts go into the score buffer's memory: with the term matrix resident there is no room for another 25 GB)
    torch.cuda.synchronize()
    panel.roh_coverage_fused_device(W, ERROR, MAX_GAP, 2.5, out.data_ptr(), pitch_align=8, use_gl=True)
    tg = []
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        panel.roh_coverage_fused_device(W, ERROR, MAX_GAP, 2.5, out.data_ptr(), pitch_align=8, use_gl=True)
        tg.append(time.perf_counter() - t0)
    res["tgls_coverage_fused"] = {"call_ms": float(np.min(tg)) * 1e3, "scores_then_counts_ms": res["tgls"]["kernel_ms"] + tcv * 1e3,
                                  "note": "lod_chain_ring_kernel writing coverage bits (4 B per individual and 32 windows) + "
                                          "cov_counts_from_bits_kernel, against the score pass + roh_coverage_kernel"}
    clk = clock_under_load(lambda: panel.wlod_windows_device(out.data_ptr(), W, ERROR, MAX_GAP, M_GEN, MU, use_gl=True),
                           torch.cuda.synchronize)
    res["wlod_gl"] = dict(rate(k, dt), roofline=fp64_roofline("wlod_strip_gl_kernel", win, W, k, clock=clk))
    st = panel.stats()
    res["liveness"] = {"n_stall_reruns": int(st["n_stall_reruns"]), "n_count_timeouts": int(st["n_count_timeouts"]),
                       "tgls_mode": list(panel.tgls_mode()),
                       "note": "strip launches the tile form had to repair / count items that gave up: expected 0 (garlic_call_stats)"}
    # GARLIC's default --winsize 10 (windows narrower than the kernels' 16-window groups): bound by the scores written
    W10 = 10
    panel.compute_ld(W10, want_output=False)                     # scratch for this window size
    ts10 = []
    for _ in range(max(2, steps // 2)):
        t0 = time.perf_counter()
        panel.compute_ld(W10, want_output=False)
        ts10.append(time.perf_counter() - t0)
    ld10 = float(np.mean(ts10))
    # an LD call at W = 10 is byte movement: packed genotypes in, bit planes out and in again, pair counts, weights
    nblk10 = (nind + 63) // 64
    ld10_bytes = (0.25 * nloci * nblk10 * 64 + 3 * 16.0 * nloci * nblk10 + 2 * 8.0 * nloci * W10 + 3 * 8.0 * nloci * W10)
    res["ld_winsize10"] = {"call_ms": ld10 * 1e3, "snps_per_s": nloci / ld10,
                           "roofline": {"bound": "hbm", "kernel": "ld_planes_kernel + ld_pair_lane_kernel + ld_sum_flat_kernel + skew (the whole warm call)",
                                        "achieved": ld10_bytes / ld10 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": ld10_bytes / ld10 / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                        "algorithmic_bytes_per_call": ld10_bytes,
                                        "note": "genotypes once (0.25 B), planes written once and read by the pair counts (+ re-staged), "
                                                "pair counts written and read, LD and weights written, LD read by the skew pass"}}
    dt, k = timed_passes(ctx, lambda: panel.wlod_windows_device(out.data_ptr(), W10, ERROR, MAX_GAP, M_GEN, MU), steps, 1,
                         torch.cuda.synchronize)
    res["wlod_winsize10"] = dict(rate(k, dt), ld_call_ms=ld10 * 1e3,
                                 roofline=hbm_roofline("wlod_stream_small_kernel", BYTES_LOD * win, k,
                                                       note="2 * 10 flops per window: the 8 B of score per window bound it"))
    res["wlod_winsize10"]["lod_windows_per_s"] = win / W10 / (dt / steps)
    # ... and with per-genotype likelihoods (round 2's tile kernel: the scaled term matrix is its second stream)
    dt, k = timed_passes(ctx, lambda: panel.wlod_windows_device(out.data_ptr(), W10, ERROR, MAX_GAP, M_GEN, MU, use_gl=True), steps, 1,
                         torch.cuda.synchronize)
    res["wlod_gl_winsize10"] = dict(rate(k, dt), roofline=hbm_roofline("wlod_stream_small_gl_kernel", BYTES_TGLS * win, k,
                                                                      note="8 B of scaled terms in, 8 B of score out per window"))
    panel.close()
    out.free()
    torch.cuda.empty_cache()
    ctx.trim()
    return res


def leg_end_to_end(ctx, dev):
    """what a drop-in calcLODWindows pays at C2 size: host int16 genotypes in (upload + 2-bit packing), one
    garlic_lod_windows call with HOST output in the reference's dense rows, PCIe both ways; and the feed-only
    form (scores + thinning on the device, 8/W bytes per window back)"""
    from garlic_amd import abi, synth
    nloci, nind, W = WORKLOADS["c2"][:3]
    spec = synth.PanelSpec(nloci, seed=20260101 + 1, max_gap=MAX_GAP)
    geno = np.empty((nloci, nind), dtype=np.int16)
    for l0, g in synth.genotype_chunks(spec, nind, dev):
        geno[l0:l0 + g.shape[0]] = g.cpu().numpy()
    t0 = time.perf_counter()
    panel = abi.Panel(ctx, spec.chr_nloci, nind)
    panel.set_map(spec.pos, spec.centro_start, spec.centro_end)
    panel.set_freq(spec.freq)
    panel.set_genotypes(geno)
    t_up = time.perf_counter() - t0
    base, pitch, total = panel.out_layout(1, nind)
    out = np.empty(total, dtype=np.float64)
    out[::512] = 0   # touch the pages once: the timed call should not pay first-touch faults
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        abi.check(abi.lib().garlic_lod_windows(panel.handle, W, ERROR, MAX_GAP, 0, 0, nind, 1, abi._vp(out.ctypes.data), abi.HOST))
        times.append(time.perf_counter() - t0)
    st = panel.stats()
    panel.lod_feed(W, ERROR, MAX_GAP, W)
    t0 = time.perf_counter()
    feed, _ = panel.lod_feed(W, ERROR, MAX_GAP, W, copy=False)
    t_feed = time.perf_counter() - t0
    panel.close()
    win = nloci * nind
    return {"workload": "C2 through the C ABI with HOST buffers (the drop-in call): int16 genotypes in, doubles out",
            "upload_and_pack_s": t_up, "lod_windows_host_output_s": min(times), "of_which_kernel_ms": st["chain_kernel_ms"],
            "value": win / W / min(times), "unit": "LOD-windows/s", "includes": "PCIe D2H of 8 GB of scores (upload timed separately)",
            "with_upload": win / W / (min(times) + t_up),
            "feed_only": {"call_s": t_feed, "value": win / W / t_feed, "unit": "LOD-windows/s", "feed_values": int(feed.shape[0])}}


def self_launch(n):
    """python bench.py --gpus N without torch.distributed.run: the same command under the launcher, as a child"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: no launcher (WORLD_SIZE unset), starting: " + " ".join(cmd), file=sys.stderr)
    return subprocess.run(cmd).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="lod", choices=["lod", "wlod", "tgls"])
    ap.add_argument("--inds", type=int, default=0, help="individuals per GPU (default: workload's)")
    ap.add_argument("--cpu-inds", type=int, default=512, help="individuals in the CPU-baseline sample")
    ap.add_argument("--cpu-inds-multi", type=int, default=16,
                    help="N > 1: individuals of its own shard every rank checks against the CPU path")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--out-candidates", type=int, default=8,
                    help="score buffers to try for the timed passes (placement in VRAM changes the kernel time)")
    ap.add_argument("--also", default="auto", help="auto | none | comma list of c3,ns,e2e (N = 1 only)")
    ap.add_argument("--also-budget-s", type=float, default=330.0,
                    help="no further `also` leg is started once the run has taken this long")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # asked for N GPUs without a launcher: start one rank per GPU as a CHILD process (nothing has touched the GPU
        # yet in this one) and relay its line -- never fall through to a one-GPU run that calls itself N
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    from garlic_amd import abi, shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    # Rehearsal of the multi-rank path on a one-GPU box (not a measurement): GARLIC_BENCH_REHEARSE=1
    # puts every rank on cuda:0 and uses gloo for the barrier / max-over-ranks (RCCL refuses two
    # ranks on one device).
    rehearse = os.environ.get("GARLIC_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    wl = args.workload or ("c2" if world == 1 else "ns")
    nloci, nind, W, seed_off, desc = WORKLOADS[wl]
    if args.inds:
        nind = args.inds
    spec = synth.PanelSpec(nloci, seed=20260101 + seed_off, max_gap=MAX_GAP)

    ctx = abi.Context(local_rank)
    ctx.set_async(True)   # repeated passes with device-resident output are enqueued back to back
    # CPU baseline: N = 1: the bounded sample of SURVEY 8(d) on rank 0.  N > 1: EVERY rank keeps a few of its own
    # individuals on the host and checks their scores bit for bit against the CPU path after the timed region (a
    # few seconds: the others wait for nobody longer than that); rank 0's timing of it is the line's cpu_baseline
    if args.no_cpu or args.mode != "lod":
        n_cpu = 0
    elif world > 1:
        n_cpu = min(args.cpu_inds_multi, nind)
    else:
        n_cpu = min(args.cpu_inds, nind)
    panel, geno_sample = load_panel(ctx, spec, nind, dev, ind_offset=rank * nind, n_cpu=n_cpu, gq=(args.mode == "tgls"))

    PITCH_ALIGN = 32
    base, pitch, total = panel.out_layout(PITCH_ALIGN, nind)
    # Where the score buffer lands in VRAM changes the chain kernel's time by up to 18 % (same code, same virtual
    # layout, same box: 1.34 .. 1.68 ms at C2 -- DESIGN.md section 4, "placement").  The library handles that for the
    # buffers it is asked for: garlic_panel_alloc_scores allocates a few candidates, times the real kernel into each
    # and keeps the fastest (its own host-output scratch is chosen the same way), which is what this bench -- like any
    # caller that follows INTEGRATION.md -- uses.  Every candidate's time is reported, with the fractions of the
    # median and the worst candidate and of one plain hipMalloc buffer beside the headline.
    placement = None
    n_cand = 1 if (args.out_candidates <= 1 or total * 8 * args.out_candidates > (96 << 30) or args.mode != "lod") else args.out_candidates

    def three_passes(ptr):
        for _ in range(2):
            panel.lod_windows_device(ptr, W, ERROR, MAX_GAP, pitch_align=PITCH_ALIGN)
        ctx.synchronize()
        for _ in range(3):
            panel.lod_windows_device(ptr, W, ERROR, MAX_GAP, pitch_align=PITCH_ALIGN)
        return float(np.mean(ctx.recent_kernel_ms(3)))

    if n_cand > 1:
        # (one plain hipMalloc buffer first, for comparison; nothing is allocated or released between the library's choice
        # and the timed region)
        plain = torch.empty(total, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        plain_ms = three_passes(plain.data_ptr())
        del plain
        torch.cuda.empty_cache()
        ctx.set_async(False)
        out_buf, times = panel.alloc_scores(W, ERROR, MAX_GAP, pitch_align=PITCH_ALIGN, nind_out=nind, candidates=n_cand)
        ctx.set_async(True)
        out = out_buf.tensor()
        placement = {"candidates_kernel_ms": times, "kept_ms": float(min(times)),
                     "median_ms": float(np.median(times)), "worst_ms": float(max(times)),
                     "one_plain_hipmalloc_buffer_ms": plain_ms,
                     "drawn": panel.alloc_scores_info(),
                     "note": "garlic_panel_alloc_scores: candidates from the library's pooled allocator side by side, the real "
                             "kernel timed into each (passes enqueued back to back, the last three of five), fastest kept -- the buffer the "
                             "timed region writes; rounds of candidates are drawn until one takes its bytes at 0.74 of the HBM peak or three "
                             "rounds / 2 s are spent (`drawn`; candidates_kernel_ms: the round the kept buffer came from)"}
    else:
        out_buf = ctx.alloc_scores(total)
        out = out_buf.tensor()
    setup = {}
    if args.mode == "wlod":
        # LD weights (calcLDData) from --ld-subsample 500 of the WHOLE panel: every rank counts over its own
        # individuals, the integer counts are summed over ranks (the one collective of the weighted path,
        # RCCL all-reduce), every rank finishes the floating-point part identically
        sub = np.sort(np.random.default_rng(20260101).choice(nind * world, size=min(500, nind * world), replace=False))
        mine = shard.split_subsample(sub, nind * world, world, rank)
        loc = torch.zeros((nloci, 2), dtype=torch.int32, device=dev)
        pair = torch.zeros((nloci, W, 2), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        panel.ld_counts_device(W, loc.data_ptr(), pair.data_ptr(), sub_idx=mine)
        if world > 1:
            if rehearse:
                l2, p2 = shard.allreduce_ld_counts(loc.cpu(), pair.cpu())
                loc.copy_(l2); pair.copy_(p2)
            else:
                shard.allreduce_ld_counts(loc, pair)
        torch.cuda.synchronize()
        panel.ld_finish_device(W, loc.data_ptr(), pair.data_ptr())
        setup["ld_weights_s"] = time.perf_counter() - t0
        del loc, pair
        torch.cuda.empty_cache()
        ctx.trim()

    def step():
        if args.mode == "lod":
            panel.lod_windows_device(out.data_ptr(), W, ERROR, MAX_GAP, pitch_align=PITCH_ALIGN)
        elif args.mode == "tgls":
            panel.lod_windows_device(out.data_ptr(), W, ERROR, MAX_GAP, pitch_align=PITCH_ALIGN, use_gl=True)
        else:
            panel.wlod_windows_device(out.data_ptr(), W, ERROR, MAX_GAP, M_GEN, MU, pitch_align=PITCH_ALIGN)

    # The first passes after device memory has been mapped or unmapped (the panel upload, the score candidates, buffers
    # given back) run slower and converge over some ten passes (1.42 -> 1.33 ms at C2; tools/exp/placement_stick.py):
    # part of the setup, like the uploads -- passes until five in a row agree within 1.5 %, 40 at most.  The contract's
    # W warm-up steps follow.
    settle = []
    for _ in range(8):
        for _ in range(5):
            step()
        settle += ctx.recent_kernel_ms(5)
        last = settle[-5:]
        if max(last) <= 1.015 * min(last):
            break
    if os.environ.get("GARLIC_BENCH_DEBUG"):
        print("debug: settling passes, kernel ms:", [round(x, 3) for x in settle], file=sys.stderr)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    solo_elapsed = None
    if world > 1:
        # rank 0 alone first, the other GPUs idle: the same K steps of the same shard, so that the joint line can be
        # read against a one-GPU point of the SAME workload (no collective on the data path: expect joint = solo;
        # what is missing is host, power or PCIe, not the kernels)
        dist.barrier()
        if rank == 0:
            ts = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            solo_elapsed = time.perf_counter() - ts
            solo_kernel_ms = float(np.mean(ctx.recent_kernel_ms(min(args.steps, 32))))
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()  # enqueues one full pass on the context's stream (same arguments: the plan is reused)
    torch.cuda.synchronize()   # device-wide: includes the context's own stream
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    own_elapsed = elapsed
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        per_rank = [float(x.item()) / args.steps * 1e3 for x in gathered]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # HIP-event durations of the dominant kernel over the timed region, on the stream it runs on
    # (one event pair per pass, read after the region: the passes were not waited for one by one)
    kernel_ms = ctx.recent_kernel_ms(min(args.steps, 32))
    if os.environ.get("GARLIC_BENCH_DEBUG"):
        print("debug: kernel ms of the timed steps:", [round(x, 3) for x in kernel_ms], file=sys.stderr)
    st = panel.stats()
    # the bit check: this rank's first n_cpu individuals against the CPU path (reference build if present, else the port)
    cpu = None
    if n_cpu:
        host = out.cpu().numpy() if total * 8 < (6 << 30) else None
        rows = []
        for c in range(spec.nchr):
            n = int(spec.chr_nloci[c])
            if host is not None:
                blk = host[base[c]: base[c] + nind * pitch[c]].reshape(nind, pitch[c])
            else:
                blk = out[base[c]: base[c] + nind * pitch[c]].view(nind, pitch[c])[:n_cpu].cpu().numpy()
            rows.append(blk[:n_cpu, :n])
        cpu = cpu_baseline(spec, geno_sample, W, rows, seconds_budget=25.0 if world == 1 else 4.0, all_cores=(world == 1))
        del host, rows
    mism_by_rank = None
    if world > 1:
        m = torch.tensor([cpu["gpu_bit_mismatches_on_sample"] if cpu else -1, cpu["values_compared"] if cpu else 0],
                         dtype=torch.int64, device="cpu" if rehearse else dev)
        g = [torch.zeros_like(m) for _ in range(world)]
        dist.all_gather(g, m)
        mism_by_rank = [[int(x[0].item()), int(x[1].item())] for x in g]
    if rank == 0:
        windows_per_step = nloci * nind * world            # sliding windows (SNPs x inds)
        lod_windows_per_step = windows_per_step / W        # BASELINE.json unit
        k_ms = float(np.mean(kernel_ms))
        win_rank = nloci * nind
        if args.mode == "lod":
            roof = hbm_roofline("lod_chain_kernel", BYTES_LOD * win_rank, k_ms)
        elif args.mode == "tgls":
            roof = hbm_roofline("lod_chain_ring_kernel", BYTES_TGLS * win_rank, k_ms)
        else:
            roof = fp64_roofline("wlod_tile2_kernel", win_rank, W, k_ms)
        res = {
            "metric": "LOD-windows/sec (SNPs x inds / winsize)",
            "value": lod_windows_per_step * args.steps / elapsed,
            "unit": "LOD-windows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": desc, "mode": {"lod": "unweighted --error", "wlod": "--weighted (LD weights from --ld-subsample 500)",
                                           "tgls": "TGLS --gl-type GQ"}[args.mode],
                "snps": nloci, "inds_per_gpu": nind, "inds_total": nind * world, "winsize": W,
                "error": ERROR, "max_gap": MAX_GAP, "output": "full FP64 scores, individual-major",
                "sharding": "individuals across GPUs, no collective on the data path",
            },
            "sliding_windows_per_s": windows_per_step * args.steps / elapsed,
            "roofline": roof,
            "plan": {k: int(st[k]) for k in ("n_segments", "n_runs", "n_chain_items", "n_valid_windows", "n_missing")},
        }
        if per_rank is not None:
            res["ms_per_step_by_rank"] = per_rank
            solo_value = nloci * nind / W * args.steps / solo_elapsed
            res["per_gpu_value"] = res["value"] / world
            res["solo_value_rank0"] = solo_value
            res["solo_ms_per_step_rank0"] = solo_elapsed / args.steps * 1e3
            res["solo_kernel_ms_rank0"] = solo_kernel_ms
            res["efficiency"] = res["per_gpu_value"] / solo_value
            res["efficiency_note"] = ("joint per-GPU rate (max over ranks) / rank 0 running the same shard alone just before, the "
                                      "other GPUs idle; ms_per_step_by_rank shows which rank set the joint time")
            res["gpu_bit_mismatches_by_rank"] = [x[0] for x in mism_by_rank]
            res["values_compared_by_rank"] = [x[1] for x in mism_by_rank]
        if setup:
            res["setup"] = setup
        if placement:
            res["output_placement"] = placement
            if args.mode == "lod":     # the same roofline fraction had the timed region written another candidate
                alg = BYTES_LOD * win_rank
                for key, name in (("median_ms", "frac_median"), ("worst_ms", "frac_worst"),
                                  ("one_plain_hipmalloc_buffer_ms", "frac_one_plain_hipmalloc_buffer")):
                    res["roofline"][name] = alg / (placement[key] * 1e-3) / 1e9 / res["roofline"]["peak"]
                res["roofline"]["placement_note"] = ("frac: the buffer garlic_panel_alloc_scores kept (what a caller of the library "
                                                     "gets); frac_median / frac_worst: the median / worst of its candidates")
        res["cpu_baseline"] = cpu
        if cpu:
            res["speedup_vs_cpu_baseline"] = res["value"] / cpu["value"]
    panel.close()
    del out, geno_sample
    if out_buf is not None:
        out_buf.free()
    torch.cuda.empty_cache()
    ctx.trim()

    if rank == 0 and world == 1 and args.also != "none":
        want = ["e2e", "ns", "c3"] if args.also == "auto" else [x for x in args.also.split(",") if x]
        also = {}
        ctx.set_async(True)
        for name in want:
            used = time.perf_counter() - T_START
            if used > args.also_budget_s:
                also[name] = {"skipped": f"time budget: {used:.0f} s used of --also-budget-s {args.also_budget_s:.0f}"}
                continue
            t0 = time.perf_counter()
            try:
                if name == "e2e":
                    ctx.set_async(False)
                    res["end_to_end"] = leg_end_to_end(ctx, dev)
                    ctx.set_async(True)
                    ctx.trim()      # (the placement candidates of the host-output scratch: idle pooled memory)
                    continue
                also[{"ns": "c4_c5_shard", "c3": "c3_multi_winsize"}[name]] = dict(
                    (leg_ns if name == "ns" else leg_c3)(ctx, dev, max(3, min(args.steps, 10))),
                    leg_wall_s=None)
                also[{"ns": "c4_c5_shard", "c3": "c3_multi_winsize"}[name]]["leg_wall_s"] = time.perf_counter() - t0
            except Exception as e:   # a leg that fails (e.g. another process holds the memory) must not cost the headline
                also[name] = {"failed": f"{type(e).__name__}: {e}"}
        res["also"] = also
    if rank == 0:
        print(json.dumps(res))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
