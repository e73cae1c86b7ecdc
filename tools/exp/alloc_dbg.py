import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from garlic_amd import abi
from tests import oracle_lib as ol
ctx = abi.Context(0)
rng = np.random.default_rng(3)
W, mg, sizes, nind = 20, 200000, [900, 333], 70
chroms = [ol.random_panel(rng, n, nind, max_gap=mg) for n in sizes]
panel = abi.Panel(ctx, sizes, nind)
panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms])
panel.set_freq(np.concatenate([c[1] for c in chroms]))
panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
base, pitch, total = panel.out_layout(32, nind)
buf = ctx.alloc_scores(total)
print("ptr", hex(buf.ptr), "total", total)
t = buf.tensor()
print("tensor ptr", hex(t.data_ptr()), t.shape, t.dtype)
panel.lod_windows_device(buf.ptr, W, 0.001, mg, pitch_align=32)
ctx.synchronize()
torch.cuda.synchronize()
h1 = t.cpu().numpy()
h2 = buf.tensor().cpu().numpy()
ref = torch.empty(total, dtype=torch.float64, device="cuda:0")
panel.lod_windows_device(ref.data_ptr(), W, 0.001, mg, pitch_align=32)
ctx.synchronize()
h3 = ref.cpu().numpy()
print("view-before == torch buffer:", np.array_equal(h1.view(np.uint64), h3.view(np.uint64)), "view-after:", np.array_equal(h2.view(np.uint64), h3.view(np.uint64)))
bad = np.nonzero(h2.view(np.uint64) != h3.view(np.uint64))[0]
print("mismatches", bad.size, bad[:5], bad[-5:] if bad.size else None)
for it in range(4):
    buf.free()
    buf = ctx.alloc_scores(total)
    panel.lod_windows_device(buf.ptr, W, 0.001, mg, pitch_align=32)
    ctx.synchronize()
    h = buf.tensor().cpu().numpy()
    bad = np.nonzero(h.view(np.uint64) != h3.view(np.uint64))[0]
    hb = np.empty(total); 
    print("iteration", it, "ptr", hex(buf.ptr), "mismatches via torch view", bad.size, (bad[:3], bad[-3:]) if bad.size else "")
    t2 = torch.empty(total, dtype=torch.float64, device="cuda:0")
    t2.copy_(buf.tensor()); torch.cuda.synchronize()
    bad2 = np.nonzero(t2.cpu().numpy().view(np.uint64) != h3.view(np.uint64))[0]
    print("   via device copy", bad2.size)
