#!/usr/bin/env python3
"""Run-to-run determinism of the score kernels on one box: the 200k x 1000 panel of tests/test_gpu_fullsize.py, every
variant N times into the same buffer, each result compared bit for bit with the first.  Reports where a difference
sits (chromosome, individual, locus) -- the one intermittent failure of test_variants_200k_by_1000_sampled_parity
(wLOD, chromosome 6) was seen once in the full suite and never alone."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from garlic_amd import abi, synth
N = int(os.environ.get("REPS", 200))
nloci, nind, W, mg = 200000, 1000, int(os.environ.get('WIN', 100)), 200000
dev = torch.device("cuda:0")
spec = synth.PanelSpec(nloci, seed=20260104, max_gap=mg)
gen = torch.Generator(device=dev); gen.manual_seed(5)
with abi.Context(0) as ctx, abi.Panel(ctx, spec.chr_nloci, nind) as panel:
    panel.set_map(spec.pos, spec.centro_start, spec.centro_end, gpos=spec.gpos)
    panel.set_freq(spec.freq)
    for l0, g in synth.genotype_chunks(spec, nind, dev):
        gq = torch.randint(3, 61, g.shape, generator=gen, device=dev).to(torch.float64)
        gl = torch.pow(torch.tensor(10.0, dtype=torch.float64, device=dev), -gq / 10.0)
        torch.cuda.synchronize()
        panel.set_genotypes_device(g.data_ptr(), g.shape[1], l0, g.shape[0])
        panel.set_gl_device(gl.data_ptr(), gl.shape[1], l0, gl.shape[0])
    panel.compute_ld(W, sub_idx=np.arange(0, nind, 10, dtype=np.int32))
    base, pitch, total = panel.out_layout(32, nind)
    runs = {
        "lod": lambda o: panel.lod_windows_device(o.data_ptr(), W, 0.001, mg),
        "wlod": lambda o: panel.wlod_windows_device(o.data_ptr(), W, 0.001, mg, 7, 1e-9),
        "tgls": lambda o: panel.lod_windows_device(o.data_ptr(), W, 0.001, mg, use_gl=True),
        "wlodgl": lambda o: panel.wlod_windows_device(o.data_ptr(), W, 0.001, mg, 7, 1e-9, use_gl=True),
    }
    res = {}
    only = os.environ.get("MODES")
    for name, call in runs.items():
        if only and name not in only.split(","):
            continue
        bad = []
        first = None
        for rep in range(N):
            out = torch.empty(total, dtype=torch.float64, device=dev) if rep % 10 == 0 else out   # a fresh buffer now and then
            out.fill_(float("nan"))
            torch.cuda.synchronize()
            call(out)
            torch.cuda.synchronize()
            if first is None:
                first = out.clone()
                continue
            neq = (out.view(torch.int64) != first.view(torch.int64)).nonzero().flatten()
            if neq.numel():
                where = []
                for off in neq[:5].tolist():
                    c = max(i for i in range(spec.nchr) if base[i] <= off)
                    r, l = divmod(off - base[c], pitch[c])
                    where.append((int(c), int(r), int(l), float(out[off]), float(first[off])))
                offs = neq.tolist()
                c0 = max(i for i in range(spec.nchr) if base[i] <= offs[0])
                rr = sorted({(o - base[c0]) // pitch[c0] for o in offs}); ll = sorted({(o - base[c0]) % pitch[c0] for o in offs})
                bad.append({"rep": rep, "n": int(neq.numel()), "rows": [int(rr[0]), int(rr[-1]), len(rr)],
                            "loci": [int(ll[0]), int(ll[-1]), len(ll)], "where": where[:2]})
        res[name] = {"reps": N, "mismatching_reps": len(bad), "first": bad[:3]}
    print(json.dumps(res))
