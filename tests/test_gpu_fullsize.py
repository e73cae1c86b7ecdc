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


def test_variants_200k_by_1000_sampled_parity(gpu_ctx):
    """wLOD (LD weights computed on the device from a 100-individual subsample) and TGLS at a size
    where every kernel runs many workgroups per CU: sampled individuals bit for bit, mask and
    idempotence over the whole output"""
    import torch
    nloci, nind, W, mg = 200_000, 1000, 100, 200000
    spec = synth.PanelSpec(nloci, seed=20260105, max_gap=mg)
    dev = torch.device("cuda", 0)
    sample = [0, 63, 64, 517, 960, 999]
    geno_s = np.empty((nloci, len(sample)), dtype=np.int16)
    gl_s = np.empty((nloci, len(sample)), dtype=np.float64)
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    with abi.Panel(gpu_ctx, spec.chr_nloci, nind) as panel:
        panel.set_map(spec.pos, spec.centro_start, spec.centro_end, gpos=spec.gpos)
        panel.set_freq(spec.freq)
        for l0, g in synth.genotype_chunks(spec, nind, dev):
            gq = torch.randint(3, 61, g.shape, generator=gen, device=dev).to(torch.float64)
            gl = torch.pow(torch.tensor(10.0, dtype=torch.float64, device=dev), -gq / 10.0)
            torch.cuda.synchronize()
            panel.set_genotypes_device(g.data_ptr(), g.shape[1], l0, g.shape[0])
            panel.set_gl_device(gl.data_ptr(), gl.shape[1], l0, gl.shape[0])
            geno_s[l0:l0 + g.shape[0]] = g[:, sample].cpu().numpy()
            gl_s[l0:l0 + g.shape[0]] = gl[:, sample].cpu().numpy()
        sub = np.arange(0, nind, 10, dtype=np.int32)
        ld = panel.compute_ld(W, sub_idx=sub)
        assert np.isfinite(ld).all() and (ld[: int(spec.chr_nloci[0]) - W + 1] >= 1.0).all()
        base, pitch, total = panel.out_layout(32, nind)
        out = torch.empty(total, dtype=torch.float64, device=dev)
        runs = {
            "wlod": lambda: panel.wlod_windows_device(out.data_ptr(), W, 0.001, mg, 7, 1e-9),
            "tgls": lambda: panel.lod_windows_device(out.data_ptr(), W, 0.001, mg, use_gl=True),
        }
        for name, call in runs.items():
            out.fill_(float("nan"))
            torch.cuda.synchronize()
            call()
            torch.cuda.synchronize()
            first = out.clone()
            out.fill_(float("nan"))
            torch.cuda.synchronize()
            call()
            torch.cuda.synchronize()
            for c in range(spec.nchr):
                n = int(spec.chr_nloci[c])
                lo, hi = int(spec.chr_off[c]), int(spec.chr_off[c + 1])
                blk = out[base[c]: base[c] + nind * pitch[c]].view(nind, pitch[c])[:, :n]
                ref = first[base[c]: base[c] + nind * pitch[c]].view(nind, pitch[c])[:, :n]
                assert torch.equal(blk.contiguous().view(torch.int64), ref.contiguous().view(torch.int64)), (name, c)
                valid = ol.oracle_mask(spec.pos[lo:hi], int(spec.centro_start[c]), int(spec.centro_end[c]), W, mg)
                miss = blk == ol.MISSING
                assert bool((miss == miss[0:1]).all()) and np.array_equal(~miss[0].cpu().numpy(), valid.astype(bool))
                g = np.ascontiguousarray(geno_s[lo:hi])
                args = (spec.freq[lo:hi], spec.pos[lo:hi])
                cen = (int(spec.centro_start[c]), int(spec.centro_end[c]))
                if name == "wlod":
                    want = ol.oracle_calc_wlod(g, *args, spec.gpos[lo:hi], ld[lo:hi], *cen, W, 0.001, mg, 1e-9, 7)
                else:
                    want = ol.oracle_calc_lod(g, *args, *cen, W, 0.001, mg, gl=np.ascontiguousarray(gl_s[lo:hi]))
                assert ol.bits_equal(blk[sample].cpu().numpy(), want), (name, c)


def test_run_to_run_determinism_200k_by_1000(gpu_ctx):
    """every score kernel, many launches into one buffer, each result bit for bit the first one.  (The two-block wLOD
    loop once waited for its look-ahead genotype word with a COUNT of the vector loads issued behind it; with more than
    63 of them outstanding the wave's VM counter let a stale word through about once in a hundred launches, 2048 wrong
    scores each time -- and tests that look at one launch pass 99 times in a hundred.)"""
    import torch
    nloci, nind, W, mg = 200_000, 1000, 100, 200000
    spec = synth.PanelSpec(nloci, seed=20260106, max_gap=mg)
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(13)
    with abi.Panel(gpu_ctx, spec.chr_nloci, nind) as panel:
        panel.set_map(spec.pos, spec.centro_start, spec.centro_end, gpos=spec.gpos)
        panel.set_freq(spec.freq)
        for l0, g in synth.genotype_chunks(spec, nind, dev):
            gq = torch.randint(3, 61, g.shape, generator=gen, device=dev).to(torch.float64)
            gl = torch.pow(torch.tensor(10.0, dtype=torch.float64, device=dev), -gq / 10.0)
            torch.cuda.synchronize()
            panel.set_genotypes_device(g.data_ptr(), g.shape[1], l0, g.shape[0])
            panel.set_gl_device(gl.data_ptr(), gl.shape[1], l0, gl.shape[0])
        panel.compute_ld(W, sub_idx=np.arange(0, nind, 10, dtype=np.int32), want_output=False)
        base, pitch, total = panel.out_layout(32, nind)
        out = torch.empty(total, dtype=torch.float64, device=dev)
        runs = {
            "wlod": (150, lambda: panel.wlod_windows_device(out.data_ptr(), W, 0.001, mg, 7, 1e-9)),
            "wlod_gl": (100, lambda: panel.wlod_windows_device(out.data_ptr(), W, 0.001, mg, 7, 1e-9, use_gl=True)),
            "lod": (60, lambda: panel.lod_windows_device(out.data_ptr(), W, 0.001, mg)),
            "tgls": (60, lambda: panel.lod_windows_device(out.data_ptr(), W, 0.001, mg, use_gl=True)),
        }
        for name, (reps, call) in runs.items():
            first = None
            for rep in range(reps):
                out.fill_(float("nan"))
                torch.cuda.synchronize()
                call()
                torch.cuda.synchronize()
                if first is None:
                    first = out.clone()
                else:
                    assert torch.equal(out.view(torch.int64), first.view(torch.int64)), (name, rep)
        # the chains that leave samples / bits instead of scores (feed_kernel.hpp; round 4: the raw term rows, the pair
        # tables and the genotype words all come through LDS behind counted waits and one barrier per tile)
        for step in (W, 7):
            feeds = [panel.lod_feed(W, 0.001, mg, step)[0] for _ in range(40)]
            assert all(ol.bits_equal(f, feeds[0]) for f in feeds[1:]), step
        _, _, t8 = panel.out_layout(8, nind)
        cov = torch.empty(t8, dtype=torch.int16, device=dev)
        first = None
        for rep in range(60):
            cov.fill_(-1)
            torch.cuda.synchronize()
            panel.roh_coverage_fused_device(W, 0.001, mg, 2.5, cov.data_ptr(), pitch_align=8)
            torch.cuda.synchronize()
            if first is None:
                first = cov.clone()
            else:
                assert torch.equal(cov, first), ("coverage from bits", rep)
        segs = [panel.roh_segments(W, 0.001, mg, 2.5, 0.25) for _ in range(40)]
        assert segs[0].shape[0] > 0 and all(np.array_equal(x, segs[0]) for x in segs[1:])


@pytest.mark.parametrize("W,step", [(100, 100), (50, 50), (100, 7)])
def test_c2_thinned_feed_equals_feed_of_full_scores(c2_panel, W, step):
    """at C2 size (1M SNPs x 1000 individuals): the feed the chain kernel thins itself
    (garlic_lod_feed) == convertWinData2DoubleData of the full scores (garlic_lod_flatten on the
    device), value for value -- two independent write-out paths of the same chain"""
    import torch
    spec, panel, sample, geno_s = c2_panel
    nind = 1000
    base, pitch, total = panel.out_layout(32, nind)
    dev = torch.device("cuda", 0)
    out = torch.empty((total,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    panel.lod_windows_device(out.data_ptr(), W, 0.001, 200000, pitch_align=32)
    cap = int(sum((int(n) + step - 1) // step for n in spec.chr_nloci)) * nind
    feed_dev = torch.empty((cap,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    n = panel.flatten_device(out.data_ptr(), step, feed_dev.data_ptr(), cap)
    torch.cuda.synchronize()
    want = feed_dev[:n].cpu().numpy()
    got, per_chr = panel.lod_feed(W, 0.001, 200000, step, copy=False)
    assert got.shape[0] == n == int(per_chr.sum())
    assert ol.bits_equal(got, want)
