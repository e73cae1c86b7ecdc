"""GPU, BASELINE.json sizes: properties that do not need the oracle over the whole output
(idempotence, the position-only MISSING mask, row sums of the mask) plus bit-exact oracle checks on a
sample of individuals spread over the 64-individual blocks (first, middle, the partial last one)."""
import numpy as np
import pytest

import oracle_lib as ol
from garlic_amd import abi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2_panel(gpu_ctx):
    import torch
    nloci, nind = 1_000_000, 1000
    spec = synth.PanelSpec(nloci, seed=20260102)
    dev = torch.device("cuda", 0)
    panel = abi.Panel(gpu_ctx, spec.chr_nloci, nind)
    panel.set_map(spec.pos, spec.centro_start, spec.centro_end, gpos=spec.gpos)
    panel.set_freq(spec.freq)
    sample = [0, 1, 63, 64, 500, 959, 960, 999]
    geno_s = np.empty((nloci, len(sample)), dtype=np.int16)
    for l0, g in synth.genotype_chunks(spec, nind, dev):
        torch.cuda.synchronize()
        panel.set_genotypes_device(g.data_ptr(), g.shape[1], l0, g.shape[0])
        geno_s[l0:l0 + g.shape[0]] = g[:, sample].cpu().numpy()
    yield spec, panel, sample, geno_s
    panel.close()


@pytest.mark.parametrize("W", [100, 50, 300])
def test_c2_properties_and_sampled_parity(c2_panel, W):
    import torch
    spec, panel, sample, geno_s = c2_panel
    nind = 1000
    base, pitch, total = panel.out_layout(32, nind)
    dev = torch.device("cuda", 0)
    out = torch.full((total,), float("nan"), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()   # the context has its own stream: order torch's fill before the call
    panel.lod_windows_device(out.data_ptr(), W, 0.001, 200000, pitch_align=32)
    torch.cuda.synchronize()
    first = out.clone()
    st = panel.stats()
    # idempotence: a second pass writes the very same bits everywhere
    out.fill_(float("nan"))
    torch.cuda.synchronize()
    panel.lod_windows_device(out.data_ptr(), W, 0.001, 200000, pitch_align=32)
    torch.cuda.synchronize()
    assert torch.equal(first.view(torch.int64), out.view(torch.int64))
    n_missing = 0
    for c in range(spec.nchr):
        n = int(spec.chr_nloci[c])
        lo, hi = int(spec.chr_off[c]), int(spec.chr_off[c + 1])
        blk = out[base[c]: base[c] + nind * pitch[c]].view(nind, pitch[c])[:, :n]
        assert not torch.isnan(blk).any()                      # every element was written
        miss = blk == ol.MISSING
        # the mask depends on positions only: identical for all individuals, equals the oracle's
        assert bool((miss == miss[0:1]).all())
        valid = ol.oracle_mask(spec.pos[lo:hi], int(spec.centro_start[c]), int(spec.centro_end[c]), W, 200000)
        assert np.array_equal(~miss[0].cpu().numpy(), valid.astype(bool))
        n_missing += int(miss[0].sum())
        # sampled individuals, bit for bit
        want = ol.oracle_calc_lod(np.ascontiguousarray(geno_s[lo:hi]), spec.freq[lo:hi], spec.pos[lo:hi],
                                  int(spec.centro_start[c]), int(spec.centro_end[c]), W, 0.001, 200000)
        got = blk[sample].cpu().numpy()
        assert ol.bits_equal(got, want), (W, c)
    assert n_missing == st["n_missing"] and st["n_valid_windows"] + st["n_missing"] == spec.nloci
