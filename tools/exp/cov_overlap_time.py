"""garlic_roh_coverage_fused, unweighted: the counts inside the chain kernel's queue against a launch of their own
(GARLIC_COVERAGE_OVERLAP=1), same panel, same process; counts compared."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from garlic_amd import abi, synth
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
nloci, nind, W = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 1250, 100
spec = synth.PanelSpec(nloci, seed=20260101 + 3, max_gap=200000)
ctx = abi.Context(0)
panel, _ = bench.load_panel(ctx, spec, nind, dev)
b8, p8, t8 = panel.out_layout(8, nind)
res = {}
for name, env in (("overlap", "1"), ("own_launch", None), ("overlap_again", "1")):
    if env: os.environ["GARLIC_COVERAGE_OVERLAP"] = env
    else: os.environ.pop("GARLIC_COVERAGE_OVERLAP", None)
    cov = torch.zeros(t8, dtype=torch.int16, device=dev)
    ts = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        panel.roh_coverage_fused_device(W, 0.001, 200000, 2.5, cov.data_ptr(), pitch_align=8)
        torch.cuda.synchronize()
        ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    res[name] = cov
    print(name, "call ms", ts, "kernels ms", round(panel.stats()["total_ms"], 3) if "total_ms" in panel.stats() else None, flush=True)
print("equal:", bool(torch.equal(res["overlap"], res["own_launch"])), bool(torch.equal(res["overlap_again"], res["own_launch"])),
      "sum", int(res["own_launch"].to(torch.int64).sum()))
