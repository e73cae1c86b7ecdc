import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import oracle_lib as ol
from garlic_amd import abi
from test_gpu_parity import make_multichr, run_gpu
W = int(sys.argv[1]); pa = int(sys.argv[2])
rng = np.random.default_rng(100 + W)
sizes = [2000, 1, W - 1 if W > 2 else 1, W, W + 1, 777, 1500]
mg = 200000
data = make_multichr(rng, sizes, 16, mg)
ctx = abi.Context(0)
out, st = run_gpu(ctx, *data, W, 0.001, mg, pitch_align=pa)
print(st)
genos, freqs, poss, css, ces = data
for c, g in enumerate(genos):
    want = ol.oracle_calc_lod(g, freqs[c], poss[c], css[c], ces[c], W, 0.001, mg)
    got = np.ascontiguousarray(out[c])
    m = got.view(np.uint64) != want.view(np.uint64)
    print("chr", c, "n", g.shape[0], "mismatch", int(m.sum()))
    if m.any():
        rows, cols = np.nonzero(m)
        print(" rows", np.unique(rows)[:20], "cols min/max", cols.min(), cols.max())
        valid = want[0] != -9999.0
        runs = np.flatnonzero(np.diff(np.concatenate([[0], valid.astype(int), [0]])))
        print(" runs", runs.reshape(-1, 2)[:10])
        r = rows[0]
        cc = cols[rows == r]
        print(" row", r, "first bad cols", cc[:10], "got", got[r, cc[:5]], "want", want[r, cc[:5]])
        # per-column mismatch count
        cnt = m.sum(axis=0)
        print(" cols with mismatches (first 40):", np.flatnonzero(cnt)[:40])
