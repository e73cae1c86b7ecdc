#!/usr/bin/env python3
"""bench.py -- Phase-I window LOD throughput on MI355X (BASELINE.json metric).

A "step" is one full pass of the hot path (calcLODWindows, reference src/garlic-roh.cpp:279) over
a synthetic SNP x individual panel that is already resident in HBM as 2-bit packed genotypes:
segment planning, MISSING fill and the LOD chain kernel, writing every window score of every
individual (FP64, individual-major) into HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3w100|small]

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU).  Individuals
shard across ranks (weak scaling: every rank owns --inds individuals of all SNPs); there is no
collective on the data path, only the timing barrier.

Rank 0 prints ONE JSON line; `roofline` prices the dominant kernel (lod_chain_kernel) against HBM,
`cpu_baseline` times the reference's own calcLOD (oracle/_ref, built from the reference sources in
the build container) or, if that library is absent, the C port in oracle/, on a bounded sample of
the same panel on the host cores -- and the GPU scores of those individuals are compared with
it bit for bit.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (nloci, inds per GPU, winsize, BASELINE.json config it is)
    "c2": (1_000_000, 1000, 100, "synthetic 1M SNPs x 1k inds, --winsize 100 --overlap-frac 0.25, unweighted LOD"),
    "c3w100": (5_000_000, 5000, 100, "synthetic 5M SNPs x 5k inds, one window size (100) of config 3"),
    "small": (100_000, 256, 100, "smoke-sized panel (not a BASELINE config)"),
}
ALG_BYTES_PER_WINDOW = 8.25  # 0.25 B 2-bit genotype in + 8 B double out
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s


def cpu_baseline(spec, geno_sample, W, error, max_gap, gpu_rows, seconds_budget=25.0):
    """Times the CPU path on `geno_sample` (int16 [nloci][n_s]) one chromosome at a time and
    checks the GPU rows against it.  Returns the cpu_baseline JSON object."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol

    use_ref = ol.have_ref()
    n_s = geno_sample.shape[1]
    t_total = 0.0
    windows = 0
    mismatches = 0
    for c in range(spec.nchr):
        lo, hi = int(spec.chr_off[c]), int(spec.chr_off[c + 1])
        g = np.ascontiguousarray(geno_sample[lo:hi])
        args = (g, spec.freq[lo:hi], spec.pos[lo:hi], int(spec.centro_start[c]),
                int(spec.centro_end[c]), W, error, max_gap)
        t0 = time.perf_counter()
        want = ol.ref_calc_lod(*args) if use_ref else ol.oracle_calc_lod(*args)
        t_total += time.perf_counter() - t0
        windows += (hi - lo) * n_s
        mismatches += ol.count_mismatch(np.ascontiguousarray(gpu_rows[c]), want)
        if t_total > seconds_budget:
            break
    lod_windows_per_s = windows / W / t_total
    # (ii) all host cores, SURVEY 8(d): independent slices of individuals, one per core, through
    # the C port (oracle/) -- the reference itself has no threaded calcLOD.  One chromosome's worth
    # of the same sample, enough to state a rate; not part of cpu_baseline.value.
    # the GPU box gives one GPU's job a share of 16 host cores whatever the affinity mask says
    ncores = min(len(os.sched_getaffinity(0)), int(os.environ.get("GARLIC_BENCH_CORES", "16")))
    lo, hi = int(spec.chr_off[0]), int(spec.chr_off[1])
    g0 = np.ascontiguousarray(geno_sample[lo:hi])
    a0 = (g0, spec.freq[lo:hi], spec.pos[lo:hi], int(spec.centro_start[0]), int(spec.centro_end[0]),
          W, error, max_gap)
    ol.oracle_calc_lod(*a0, threads=ncores)                 # page in, spin up the team
    reps, t_all = 0, 0.0
    while t_all < 3.0 and reps < 20:
        t0 = time.perf_counter()
        ol.oracle_calc_lod(*a0, threads=ncores)
        t_all += time.perf_counter() - t0
        reps += 1
    all_cores = (hi - lo) * n_s * reps / W / t_all
    return {
        "value": lod_windows_per_s,
        "unit": "LOD-windows/s",
        "cores": 1,
        "kind": "reference" if use_ref else "port",
        "sample": f"{n_s} individuals x {windows // n_s} SNPs ({c + 1} of {spec.nchr} chromosomes) of the same panel, "
                  f"single thread (calcLOD is single-threaded in the reference), {t_total:.1f} s",
        "sliding_windows_per_s": windows / t_total,
        "gpu_bit_mismatches_on_sample": int(mismatches),
        "all_cores": {"value": all_cores, "unit": "LOD-windows/s", "cores": ncores, "kind": "port",
                      "sample": f"{n_s} individuals x {hi - lo} SNPs (chromosome 1), {n_s // max(1, ncores)} "
                                f"individuals per core, {reps} repetitions, {t_all:.1f} s"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--inds", type=int, default=0, help="individuals per GPU (default: workload's)")
    ap.add_argument("--cpu-inds", type=int, default=512, help="individuals in the CPU-baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from garlic_amd import abi, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"

    nloci, nind, W, desc = WORKLOADS[args.workload]
    if args.inds:
        nind = args.inds
    error, max_gap = 0.001, 200000
    cfg_index = {"c2": 1, "c3w100": 2, "small": 0}[args.workload]
    spec = synth.PanelSpec(nloci, seed=20260101 + cfg_index, max_gap=max_gap)

    ctx = abi.Context(local_rank)
    ctx.set_async(True)   # repeated passes with device-resident output are enqueued back to back
    panel = abi.Panel(ctx, spec.chr_nloci, nind)
    panel.set_map(spec.pos, spec.centro_start, spec.centro_end, gpos=spec.gpos)
    panel.set_freq(spec.freq)
    # CPU baseline: rank 0 of the single-GPU run only (the other ranks of a larger run would wait for it)
    n_cpu = 0 if (args.no_cpu or rank != 0 or world > 1) else min(args.cpu_inds, nind)
    geno_sample = np.empty((nloci, n_cpu), dtype=np.int16) if n_cpu else None
    for l0, g in synth.genotype_chunks(spec, nind, dev, ind_offset=rank * nind):
        torch.cuda.synchronize()
        panel.set_genotypes_device(g.data_ptr(), g.shape[1], l0, g.shape[0])
        if n_cpu:
            geno_sample[l0:l0 + g.shape[0]] = g[:, :n_cpu].cpu().numpy()
    del g

    PITCH_ALIGN = 32
    base, pitch, total = panel.out_layout(PITCH_ALIGN, nind)
    out = torch.empty(total, dtype=torch.float64, device=dev)

    def step():
        panel.lod_windows_device(out.data_ptr(), W, error, max_gap, pitch_align=PITCH_ALIGN)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()  # enqueues one full pass on the context's stream (same arguments: the plan is reused)
    torch.cuda.synchronize()   # device-wide: includes the context's own stream
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # HIP-event durations of the dominant kernel over the timed region, on the stream it runs on
    # (one event pair per pass, read after the region: the passes were not waited for one by one)
    kernel_ms = ctx.recent_kernel_ms(min(args.steps, 32))
    st = panel.stats()
    if rank == 0:
        windows_per_step = nloci * nind * world            # sliding windows (SNPs x inds)
        lod_windows_per_step = windows_per_step / W        # BASELINE.json unit
        ms_per_step = elapsed / args.steps * 1e3
        k_ms = float(np.mean(kernel_ms))
        achieved = ALG_BYTES_PER_WINDOW * nloci * nind / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath) and args.workload == "c2" and nind == WORKLOADS["c2"][1]:
            with open(tpath) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        res = {
            "metric": "LOD-windows/sec (SNPs x inds / winsize)",
            "value": lod_windows_per_step * args.steps / elapsed,
            "unit": "LOD-windows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": desc,
                "snps": nloci, "inds_per_gpu": nind, "inds_total": nind * world, "winsize": W,
                "error": error, "max_gap": max_gap, "output": "full FP64 scores, individual-major",
                "sharding": "individuals across GPUs, no collective",
            },
            "sliding_windows_per_s": windows_per_step * args.steps / elapsed,
            "roofline": {
                "bound": "hbm",
                "kernel": "lod_chain_kernel",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel_ms": k_ms,
                "algorithmic_bytes_per_launch": ALG_BYTES_PER_WINDOW * nloci * nind,
            },
            "plan": {k: int(st[k]) for k in ("n_segments", "n_runs", "n_chain_items",
                                             "n_valid_windows", "n_missing")},
        }
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
            res["cpu_baseline"] = cpu_baseline(spec, geno_sample, W, error, max_gap, rows)
            res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    panel.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
