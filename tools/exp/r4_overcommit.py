"""Does device memory held idle in the library's score-buffer pool push later allocations out of VRAM?  (One bench run of
round 4 had the likelihood legs of the 10M x 1250 shard 4-60 x slower, socket power 380 W: reads at PCIe speed.)
Fill the pool with POOL_GB of idle buffers, then the shard's panel, scores and term matrix; time the TGLS chain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from garlic_amd import abi, synth
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = abi.Context(0)
pool_gb = float(os.environ.get("POOL_GB", "70"))
def mem(tag):
    free, total = torch.cuda.mem_get_info()
    print("%-28s free %.1f GB of %.1f; library (live, pooled, reserved) GB %s" % (tag, free / 1e9, total / 1e9, tuple(round(x / 1e9, 1) for x in ctx.alloc_stats())), flush=True)
mem("start")
bufs = [ctx.alloc_scores(int(10e9 / 8)) for _ in range(int(pool_gb / 10))]
for b in bufs: b.free()
mem("pool filled")
spec = synth.PanelSpec(10_000_000, seed=20260101 + 3, max_gap=200000)
panel, _ = bench.load_panel(ctx, spec, 1250, dev, gq=True)
mem("panel + likelihood codes")
base, pitch, total = panel.out_layout(32, 1250)
out = ctx.alloc_scores(total)
mem("scores allocated")
for k in range(3):
    panel.lod_windows_device(out.data_ptr(), 100, 0.001, 200000, use_gl=True)
    torch.cuda.synchronize()
    print("tgls pass", k, "kernel ms", round(panel.stats()["chain_kernel_ms"], 2), flush=True)
mem("after tgls")
panel.compute_ld(100, want_output=False)
for k in range(2):
    panel.wlod_windows_device(out.data_ptr(), 100, 0.001, 200000, 7, 1e-9, use_gl=True)
    torch.cuda.synchronize()
    print("wlod_gl pass", k, "kernel ms", round(panel.stats()["chain_kernel_ms"], 2), "stall reruns", panel.stats()["n_stall_reruns"], flush=True)
mem("after wlod_gl")
