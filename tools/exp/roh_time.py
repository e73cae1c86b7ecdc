import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from garlic_amd import abi, synth
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
nloci, nind, W = 2_000_000, 1280, 100
spec = synth.PanelSpec(nloci, seed=3, max_gap=200000)
ctx = abi.Context(0)
panel, _ = bench.load_panel(ctx, spec, nind, dev)
base, pitch, total = panel.out_layout(32, nind)
out = torch.empty(total, dtype=torch.float64, device=dev)
panel.lod_windows_device(out.data_ptr(), W, 0.001, 200000)
ctx.synchronize()
b1, p1, t1 = panel.out_layout(1, nind)
cov = torch.empty(t1, dtype=torch.int16, device=dev)
L = abi.lib()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    abi.check(L.garlic_roh_coverage(panel.handle, C.c_void_p(out.data_ptr()), 32, nind, W, C.c_double(2.5), C.c_void_p(cov.data_ptr()), 1, abi.DEVICE))
    ctx.synchronize(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"roh_coverage device->device: {dt*1e3:.2f} ms  ({total*8/dt/1e12:.2f} TB/s of scores read)")
# the same counts without the score matrix (garlic_roh_coverage_fused): chain + compare + sliding count in one kernel
b8, p8, t8 = panel.out_layout(8, nind)
cov8 = torch.empty(t8, dtype=torch.int16, device=dev)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    panel.roh_coverage_fused_device(W, 0.001, 200000, 2.5, cov8.data_ptr(), pitch_align=8)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"roh_coverage_fused (no scores): {dt*1e3:.2f} ms call, kernel {ctx.recent_kernel_ms(1)[0]:.2f} ms  ({nloci*nind/dt/1e9:.0f} G windows/s)")
a = cov.view(torch.int16)
ok = True
for c in range(len(b1)):
    n = spec.chr_nloci[c]
    x = cov[b1[c]: b1[c] + nind * p1[c]].view(nind, p1[c])[:, :n]
    y = cov8[b8[c]: b8[c] + nind * p8[c]].view(nind, p8[c])[:, :n]
    ok = ok and bool(torch.equal(x, y))
print("fused == scores-then-counts:", ok)
