"""GPU parity: libgarlic_hip (through the C ABI) vs the CPU oracle, bit for bit."""
import numpy as np
import pytest

import oracle_lib as ol
from garlic_amd import abi

pytestmark = pytest.mark.gpu


def make_multichr(rng, chr_sizes, nind, max_gap, **kw):
    genos, freqs, poss, css, ces = [], [], [], [], []
    for n in chr_sizes:
        g, f, p, cs, ce = ol.random_panel(rng, n, nind, max_gap=max_gap, **kw)
        genos.append(g); freqs.append(f); poss.append(p); css.append(cs); ces.append(ce)
    return genos, freqs, poss, css, ces


def run_gpu(ctx, genos, freqs, poss, css, ces, W, error, max_gap, pitch_align=1, ind_begin=0,
            ind_count=None):
    nind = genos[0].shape[1]
    with abi.Panel(ctx, [g.shape[0] for g in genos], nind) as panel:
        panel.set_map(np.concatenate(poss), css, ces)
        panel.set_freq(np.concatenate(freqs))
        panel.set_genotypes(np.concatenate(genos, axis=0))
        out = panel.lod_windows(W, error, max_gap, pitch_align=pitch_align, ind_begin=ind_begin,
                                ind_count=ind_count)
        return out, panel.stats()


def check_against_oracle(out, genos, freqs, poss, css, ces, W, error, max_gap, lo=0, hi=None):
    for c, g in enumerate(genos):
        want = ol.oracle_calc_lod(g, freqs[c], poss[c], css[c], ces[c], W, error, max_gap)
        want = want[lo:hi]
        got = np.ascontiguousarray(out[c])
        assert got.shape == want.shape
        bad = ol.count_mismatch(got, want)
        assert bad == 0, f"chr {c}: {bad} of {want.size} doubles differ"


@pytest.mark.parametrize("W", [2, 5, 30, 60, 100, 300])
@pytest.mark.parametrize("pitch_align", [1, 32])
def test_unweighted_parity(gpu_ctx, W, pitch_align):
    rng = np.random.default_rng(100 + W)
    sizes = [2000, 1, W - 1 if W > 2 else 1, W, W + 1, 777, 1500]
    max_gap = 200000
    data = make_multichr(rng, sizes, 16, max_gap)
    out, st = run_gpu(gpu_ctx, *data, W, 0.001, max_gap, pitch_align=pitch_align)
    check_against_oracle(out, *data, W, 0.001, max_gap)
    assert st["n_valid_windows"] + st["n_missing"] == sum(sizes)


@pytest.mark.parametrize("nind", [1, 63, 64, 65, 130, 200])
def test_individual_counts(gpu_ctx, nind):
    rng = np.random.default_rng(7 + nind)
    max_gap = 50000
    data = make_multichr(rng, [900, 1300], nind, max_gap, gaps=4)
    for pa in (1, 2, 32):
        out, _ = run_gpu(gpu_ctx, *data, 25, 0.01, max_gap, pitch_align=pa)
        check_against_oracle(out, *data, 25, 0.01, max_gap)


def test_individual_subrange(gpu_ctx):
    rng = np.random.default_rng(5)
    max_gap = 200000
    data = make_multichr(rng, [1200, 800], 150, max_gap)
    out, _ = run_gpu(gpu_ctx, *data, 40, 0.001, max_gap, pitch_align=32, ind_begin=37, ind_count=70)
    check_against_oracle(out, *data, 40, 0.001, max_gap, lo=37, hi=107)
