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


@pytest.mark.parametrize("pitch_align", [1, 2, 32])
@pytest.mark.parametrize("W", [5, 100])
def test_device_output_layouts(gpu_ctx, W, pitch_align):
    """device-resident output in the caller's own layout: dense rows (pitch_align 1: the generic
    kernel path, 8-byte stores), even pitch, 256-byte rows.  Host output always computes in the
    padded layout and copies rows out, so only this test reaches the dense-layout kernel."""
    import torch
    rng = np.random.default_rng(40 + W)
    sizes = [1500, W, 700]
    max_gap = 200000
    data = make_multichr(rng, sizes, 70, max_gap)
    genos, freqs, poss, css, ces = data
    with abi.Panel(gpu_ctx, sizes, 70) as panel:
        panel.set_map(np.concatenate(poss), css, ces)
        panel.set_freq(np.concatenate(freqs))
        panel.set_genotypes(np.concatenate(genos, axis=0))
        base, pitch, total = panel.out_layout(pitch_align, 70)
        out = torch.full((total + 1,), float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        # + 1 element offset for pitch_align 1: not even 16-byte aligned
        off = 1 if pitch_align == 1 else 0
        panel.lod_windows_device(out.data_ptr() + 8 * off, W, 0.001, max_gap, pitch_align=pitch_align)
        host = out.cpu().numpy()[off:]
    for c, n in enumerate(sizes):
        got = host[base[c]: base[c] + 70 * pitch[c]].reshape(70, pitch[c])[:, :n]
        want = ol.oracle_calc_lod(genos[c], freqs[c], poss[c], css[c], ces[c], W, 0.001, max_gap)
        assert ol.count_mismatch(np.ascontiguousarray(got), want) == 0, (c, pitch_align)


@pytest.mark.parametrize("nind", [1, 63, 64, 65, 130, 200])
def test_individual_counts(gpu_ctx, nind):
    rng = np.random.default_rng(7 + nind)
    max_gap = 50000
    data = make_multichr(rng, [900, 1300], nind, max_gap, gaps=4)
    for pa in (1, 2, 32):
        out, _ = run_gpu(gpu_ctx, *data, 25, 0.01, max_gap, pitch_align=pa)
        check_against_oracle(out, *data, 25, 0.01, max_gap)


@pytest.mark.parametrize("W", [640, 990, 1010, 1025, 1040, 1100, 1600])
def test_wide_windows_around_the_genotype_ring_limit(gpu_ctx, W):
    """The tuned loop keeps the leaving SNP stream in an LDS genotype ring; windows wider than the
    ring fall back to the generic path.  Both sides of the limit (about 1000 SNPs), block-aligned
    and unaligned first individual."""
    rng = np.random.default_rng(W)
    max_gap = 10 ** 9
    data = make_multichr(rng, [5000, 3100], 140, max_gap, gaps=0)
    for ind_begin, ind_count in ((0, 140), (64, 70), (37, 64)):
        out, _ = run_gpu(gpu_ctx, *data, W, 0.001, max_gap, pitch_align=32, ind_begin=ind_begin,
                         ind_count=ind_count)
        check_against_oracle(out, *data, W, 0.001, max_gap, lo=ind_begin, hi=ind_begin + ind_count)


def test_individual_subrange(gpu_ctx):
    rng = np.random.default_rng(5)
    max_gap = 200000
    data = make_multichr(rng, [1200, 800], 150, max_gap)
    out, _ = run_gpu(gpu_ctx, *data, 40, 0.001, max_gap, pitch_align=32, ind_begin=37, ind_count=70)
    check_against_oracle(out, *data, 40, 0.001, max_gap, lo=37, hi=107)


def test_golden_unweighted(gpu_ctx):
    """The committed reference outputs (tests/golden/unweighted.npz), all chromosomes in one panel:
    >max_gap holes, a centromere that contains SNPs, an unknown chromosome (centromere 0,0),
    missing genotypes, freq in {0,1}, nloci < W and nloci == W."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "unweighted.npz"))
    nchr = int(d["nchr"])
    genos = [d[f"geno{c}"] for c in range(nchr)]
    freqs = [d[f"freq{c}"] for c in range(nchr)]
    poss = [d[f"pos{c}"] for c in range(nchr)]
    css = [int(d[f"centro{c}"][0]) for c in range(nchr)]
    ces = [int(d[f"centro{c}"][1]) for c in range(nchr)]
    for W in d["winsizes"]:
        for pa in (1, 32):
            out, _ = run_gpu(gpu_ctx, genos, freqs, poss, css, ces, int(W), float(d["error"]),
                             int(d["max_gap"]), pitch_align=pa)
            for c in range(nchr):
                assert ol.bits_equal(np.ascontiguousarray(out[c]), d[f"win{c}_W{W}"]), (int(W), pa, c)


def test_invalid_arguments(gpu_ctx):
    rng = np.random.default_rng(1)
    g, f, p, cs, ce = ol.random_panel(rng, 200, 8)
    with abi.Panel(gpu_ctx, [200], 8) as panel:
        with pytest.raises(abi.GarlicError) as e:          # inputs missing
            panel.lod_windows(10, 0.001, 200000)
        assert e.value.code == abi.ERR_STATE
        panel.set_map(p, [cs], [ce])
        panel.set_freq(f)
        panel.set_genotypes(g)
        with pytest.raises(abi.GarlicError) as e:          # winsize must be > 1 (garlic-cli.cpp:433)
            panel.lod_windows(1, 0.001, 200000)
        assert e.value.code == abi.ERR_INVALID
        with pytest.raises(abi.GarlicError):
            panel.lod_windows(10, 0.001, 200000, ind_begin=4, ind_count=8)
    with pytest.raises(abi.GarlicError):                   # initWinData refuses empty shapes
        abi.Panel(gpu_ctx, [0], 8)


def test_chunked_genotype_upload_and_repeat_calls(gpu_ctx):
    rng = np.random.default_rng(11)
    max_gap = 200000
    data = make_multichr(rng, [700, 500], 70, max_gap)
    genos, freqs, poss, css, ces = data
    allg = np.concatenate(genos, axis=0)
    with abi.Panel(gpu_ctx, [700, 500], 70) as panel:
        panel.set_map(np.concatenate(poss), css, ces)
        panel.set_freq(np.concatenate(freqs))
        for l0 in range(0, 1200, 133):                      # ragged chunks, unaligned to 16 SNPs
            panel.set_genotypes(allg[l0:l0 + 133], locus_begin=l0)
        for W, err in ((20, 0.001), (50, 0.001), (20, 0.01), (20, 0.001)):
            out = panel.lod_windows(W, err, max_gap, pitch_align=32)
            check_against_oracle(out, *data, W, err, max_gap)


def test_async_context_enqueues_repeated_calls(gpu_ctx):
    """garlic_ctx_set_async: repeated device-output calls only enqueue; after garlic_ctx_synchronize (or a
    stats read) the scores are the same bits as a synchronous call's"""
    import torch
    rng = np.random.default_rng(77)
    sizes = [3000, 1200]
    max_gap = 200000
    data = make_multichr(rng, sizes, 130, max_gap)
    genos, freqs, poss, css, ces = data
    ctx = abi.Context(0)
    ctx.set_async(True)
    with abi.Panel(ctx, sizes, 130) as panel:
        panel.set_map(np.concatenate(poss), css, ces)
        panel.set_freq(np.concatenate(freqs))
        panel.set_genotypes(np.concatenate(genos, axis=0))
        base, pitch, total = panel.out_layout(32, 130)
        out = torch.full((total,), float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        for _ in range(4):                                   # first call plans, the others enqueue
            panel.lod_windows_device(out.data_ptr(), 50, 0.001, max_gap, pitch_align=32)
        assert panel.stats()["chain_kernel_ms"] > 0          # waits for the last pass
        ctx.synchronize()
        host = out.cpu().numpy()
    ctx.close()
    for c, n in enumerate(sizes):
        got = host[base[c]: base[c] + 130 * pitch[c]].reshape(130, pitch[c])[:, :n]
        want = ol.oracle_calc_lod(genos[c], freqs[c], poss[c], css[c], ces[c], 50, 0.001, max_gap)
        assert ol.count_mismatch(np.ascontiguousarray(got), want) == 0


def test_random_configurations(gpu_ctx):
    """40 random panels: chromosome counts and lengths, window sizes, gap density, individual counts,
    sub-ranges and layouts drawn at random; every double against the oracle"""
    rng = np.random.default_rng(20261004)
    for trial in range(40):
        nchr = int(rng.integers(1, 5))
        sizes = [int(rng.integers(1, 2600)) for _ in range(nchr)]
        W = int(rng.choice([2, 3, 7, 16, 33, 64, 100, 257]))
        nind = int(rng.integers(1, 200))
        max_gap = int(rng.choice([3000, 50000, 200000]))
        data = make_multichr(rng, sizes, nind, max_gap, gaps=int(rng.integers(0, 6)), miss=float(rng.choice([0.0, 0.03, 0.3])))
        i0 = int(rng.integers(0, nind))
        cnt = int(rng.integers(1, nind - i0 + 1))
        pa = int(rng.choice([1, 2, 32]))
        err = float(rng.choice([1e-6, 0.001, 0.05, 0.5]))
        out, st = run_gpu(gpu_ctx, *data, W, err, max_gap, pitch_align=pa, ind_begin=i0, ind_count=cnt)
        genos, freqs, poss, css, ces = data
        for c, g in enumerate(genos):
            want = ol.oracle_calc_lod(g, freqs[c], poss[c], css[c], ces[c], W, err, max_gap)[i0:i0 + cnt]
            bad = ol.count_mismatch(np.ascontiguousarray(out[c]), want)
            assert bad == 0, (trial, sizes, W, nind, i0, cnt, pa, c, bad)
        assert st["n_valid_windows"] + st["n_missing"] == sum(sizes)


def pack2bit(geno):
    """[nloci][nind] int16 (-9 missing) -> SNP-major 2-bit rows, 4 genotypes per byte"""
    nloci, nind = geno.shape
    code = np.where((geno >= 0) & (geno <= 2), geno, 3).astype(np.uint8)
    pad = (-nind) % 4
    code = np.concatenate([code, np.full((nloci, pad), 3, np.uint8)], axis=1).reshape(nloci, -1, 4)
    return (code[:, :, 0] | (code[:, :, 1] << 2) | (code[:, :, 2] << 4) | (code[:, :, 3] << 6)).astype(np.uint8)


@pytest.mark.parametrize("nind,lo", [(70, 0), (45, 13), (64, 64), (1, 6)])
def test_genotypes_from_2bit_rows(gpu_ctx, nind, lo):
    """garlic_panel_set_genotypes_2bit: a shard's individuals [lo, lo + nind) of 2-bit rows that hold
    the whole data set, streamed in chunks -> the same scores as the int16 upload"""
    rng = np.random.default_rng(100 * nind + lo)
    total = lo + nind + 5
    sizes = [700, 333]
    W, mg = 25, 200000
    chroms = [ol.random_panel(rng, n, total, max_gap=mg) for n in sizes]
    geno = np.concatenate([c[0] for c in chroms], axis=0)
    rows = pack2bit(geno)
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms])
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes_2bit(rows[:400], ind_offset=lo)
        panel.set_genotypes_2bit(rows[400:], ind_offset=lo, locus_begin=400)
        got = panel.lod_windows(W, 0.001, mg)
        for c, (g, f, p, cs, ce) in enumerate(chroms):
            want = ol.oracle_calc_lod(g[:, lo:lo + nind], f, p, cs, ce, W, 0.001, mg)
            assert ol.bits_equal(got[c], want), c
        with pytest.raises(abi.GarlicError):
            panel.set_genotypes_2bit(rows[:, :(lo + nind + 3) // 4 - 1] if (lo + nind + 3) // 4 > 1 else rows[:, :0],
                                     ind_offset=lo)


def test_several_window_sizes_in_one_call(gpu_ctx):
    """garlic_lod_windows_multi (--winsize-multi): one resident panel, one score block per window size"""
    rng = np.random.default_rng(55)
    sizes, nind, mg = [900, 300], 70, 200000
    data = make_multichr(rng, sizes, nind, mg, gaps=2, miss=0.03)
    genos, freqs, poss, css, ces = data
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate(poss), css, ces)
        panel.set_freq(np.concatenate(freqs))
        panel.set_genotypes(np.concatenate(genos, axis=0))
        for pa in (1, 32):
            got = panel.lod_windows_multi([50, 10, 300, 33], 0.001, mg, pitch_align=pa)
            for W, blocks in got.items():
                for c, g in enumerate(genos):
                    want = ol.oracle_calc_lod(g, freqs[c], poss[c], css[c], ces[c], W, 0.001, mg)
                    assert ol.bits_equal(np.ascontiguousarray(blocks[c]), want), (pa, W, c)
        with pytest.raises(abi.GarlicError):
            panel.lod_windows_multi([50, 1], 0.001, mg)
