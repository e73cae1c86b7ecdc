"""GPU, at the shard shapes the north star is about (BASELINE.json configs 3-5): one GPU's share of the
10M-SNP x 10k-individual panel (10M x 1250) through every variant -- unweighted LOD, the thinned KDE feed,
LD weights from a 500-individual subsample, wLOD, TGLS (--gl-type GQ dictionary codes, and continuous
likelihoods with the values converted in place), GL-weighted wLOD -- and 5M x 5k through four window
sizes.  At these shapes one chromosome block of the output exceeds 4 GB, the TGLS term matrix strides
5 GB per 64-individual block and the LD tables are 8 GB each, so every 64-bit offset path is taken.

Checked bit for bit against the oracle on individuals sampled from the first and the last (partial)
64-individual block, from both sides of a block border and from rows whose offset inside a chromosome
block lies beyond 4 GB; plus, over the whole output, the position-only MISSING mask and that every
element was written.  The oracle needs seconds for a handful of individuals; the panels are generated on
the device (garlic_amd/synth.py)."""
import numpy as np
import pytest

import oracle_lib as ol
from garlic_amd import abi, synth

pytestmark = pytest.mark.gpu
MG, ERR = 200000, 0.001


def load_panel(ctx, spec, nind, sample, gq_seed=None, slices=()):
    """genotypes drawn on the device, chunk by chunk; the sampled individuals' columns are kept for the
    oracle, and every individual's genotypes for the SNP ranges in `slices` ((first global locus, count), ..).
    gq_seed: also GQ integers U{3..60} as likelihoods (config 5), uploaded as doubles."""
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.empty_cache()
    ctx.trim()            # score buffers idle in the library's pool: these tests need nearly all of the device memory
    kept = [np.empty((n, nind), dtype=np.int16) for _, n in slices]
    panel = abi.Panel(ctx, spec.chr_nloci, nind)
    panel.set_map(spec.pos, spec.centro_start, spec.centro_end, gpos=spec.gpos)
    panel.set_freq(spec.freq)
    geno_s = np.empty((spec.nloci, len(sample)), dtype=np.int16)
    gl_s = np.empty((spec.nloci, len(sample)), dtype=np.float64) if gq_seed is not None else None
    gen = torch.Generator(device=dev)
    if gq_seed is not None:
        gen.manual_seed(gq_seed)
    for l0, g in synth.genotype_chunks(spec, nind, dev):
        torch.cuda.synchronize()
        panel.set_genotypes_device(g.data_ptr(), g.shape[1], l0, g.shape[0])
        geno_s[l0:l0 + g.shape[0]] = g[:, sample].cpu().numpy()
        for (glo, n), dst in zip(slices, kept):
            a, b = max(glo, l0), min(glo + n, l0 + g.shape[0])
            if a < b:
                dst[a - glo:b - glo] = g[a - l0:b - l0].cpu().numpy()
        if gq_seed is not None:
            gq = torch.randint(3, 61, g.shape, generator=gen, device=dev).to(torch.float64)
            gl = torch.pow(torch.tensor(10.0, dtype=torch.float64, device=dev), -gq / 10.0)
            torch.cuda.synchronize()
            panel.set_gl_device(gl.data_ptr(), gl.shape[1], l0, gl.shape[0])
            gl_s[l0:l0 + g.shape[0]] = gl[:, sample].cpu().numpy()
            del gq, gl
    return (panel, geno_s, gl_s, kept) if slices else (panel, geno_s, gl_s)


def chrom_args(spec, c):
    lo, hi = int(spec.chr_off[c]), int(spec.chr_off[c + 1])
    return lo, hi, (spec.freq[lo:hi], spec.pos[lo:hi]), (int(spec.centro_start[c]), int(spec.centro_end[c]))


def check_whole_output(spec, out, base, pitch, nind, W, what):
    """every element written (the buffer was NaN before the call -- no variant here produces NaNs), and the
    MISSING mask is the position-only one of the oracle, identical for every individual"""
    import torch
    for c in range(spec.nchr):
        n = int(spec.chr_nloci[c])
        lo, hi, _, cen = chrom_args(spec, c)
        blk = out[base[c]: base[c] + nind * pitch[c]].view(nind, pitch[c])[:, :n]
        assert not bool(torch.isnan(blk).any()), (what, c, "unwritten elements")
        miss = blk == ol.MISSING
        assert bool((miss == miss[0:1]).all()), (what, c)
        valid = ol.oracle_mask(spec.pos[lo:hi], *cen, W, MG)
        assert np.array_equal(~miss[0].cpu().numpy(), valid.astype(bool)), (what, c)
        del blk, miss


def sampled_rows(out, base, pitch, nind, c, n, sample):
    return out[base[c]: base[c] + nind * pitch[c]].view(nind, pitch[c])[:, :n][sample].cpu().numpy()


def test_10M_by_1250_every_variant(gpu_ctx):
    import torch
    nloci, nind, W = 10_000_000, 1250, 100
    spec = synth.PanelSpec(nloci, seed=20260104, max_gap=MG)
    dev = torch.device("cuda", 0)
    # first block, both sides of a block border, rows on both sides of the 4-GB offset inside chromosome
    # 1's block (pitch ~ 6.4 MB per row -> row 640 onward), the partial last block (1216 ..)
    sample = [0, 1, 63, 64, 639, 640, 1215, 1216, 1249]
    # LD weights are checked on two 700-SNP slices: rows of windows that lie wholly inside a slice depend on
    # it alone -- the first loci of the panel and the last chromosome's end (rows beyond 4 GB of the table)
    n_last = int(spec.chr_nloci[-1])
    ld_slices = [(0, 700), (int(spec.chr_off[-2]) + n_last - 700, 700)]
    panel, geno_s, gl_s, ld_geno = load_panel(gpu_ctx, spec, nind, sample, gq_seed=5, slices=ld_slices)
    wsel = [0, 3, 5, 7, 8]                     # the weighted variants: individuals 0, 64, 640, 1216, 1249
    out = None
    try:
        base, pitch, total = panel.out_layout(32, nind)
        assert max(int(p) for p in pitch) * 8 * 640 > (1 << 32)           # the sample does straddle 4 GB
        out = torch.empty(total, dtype=torch.float64, device=dev)

        def run(call, what, want_fn, whole=True, sel=None):
            out.fill_(float("nan"))
            torch.cuda.synchronize()
            call()
            torch.cuda.synchronize()
            if whole:
                check_whole_output(spec, out, base, pitch, nind, W, what)
            rows = sample if sel is None else [sample[k] for k in sel]
            for c in range(spec.nchr):
                lo, hi, fp, cen = chrom_args(spec, c)
                got = sampled_rows(out, base, pitch, nind, c, hi - lo, rows)
                assert ol.bits_equal(got, want_fn(c, lo, hi, fp, cen)), (what, c)

        # ---- unweighted --error scores, twice (the second pass reuses the resident plan)
        lod_want = {}

        def want_lod(c, lo, hi, fp, cen):
            if c not in lod_want:
                lod_want[c] = ol.oracle_calc_lod(np.ascontiguousarray(geno_s[lo:hi]), *fp, *cen, W, ERR, MG, threads=16)
            return lod_want[c]

        run(lambda: panel.lod_windows_device(out.data_ptr(), W, ERR, MG), "lod", want_lod)
        run(lambda: panel.lod_windows_device(out.data_ptr(), W, ERR, MG), "lod again", want_lod, whole=False)

        # ---- the KDE feed of the sampled individuals (thinned by the chain kernel itself, step = W)
        feed, per_chr = panel.lod_feed(W, ERR, MG, W, ind_idx=np.array(sample))
        want = [ol.oracle_flatten_subset(lod_want[c], W, np.arange(len(sample))) for c in range(spec.nchr)]
        assert [len(w) for w in want] == list(per_chr)
        assert ol.bits_equal(feed, np.concatenate(want))
        feed_all, per_all = panel.lod_feed(W, ERR, MG, W, copy=False)     # everyone: 1.25e8 values
        assert int(per_all.sum()) == feed_all.shape[0] and feed_all.shape[0] > 1.2e8
        # the first individual's values open every chromosome's stretch of the feed
        off = 0
        for c in range(spec.nchr):
            w0 = ol.oracle_flatten(lod_want[c][:1], W)
            assert ol.bits_equal(feed_all[off:off + w0.shape[0]], w0), c
            off += int(per_all[c])

        # ---- the final pass without scores or counts: ROH segments straight from the genotypes (garlic_roh_segments);
        #      the sampled individuals' segments against the oracle's walk over the oracle's counts of the oracle's scores
        cutoff, frac = 2.5, 0.25
        segs = panel.roh_segments(W, ERR, MG, cutoff, frac)
        assert segs.shape[0] > nind and (np.diff(segs[:, 0]) >= 0).all()
        n_checked = 0
        for c in range(spec.nchr):
            lo, hi, fp, cen = chrom_args(spec, c)
            cov = ol.oracle_roh_coverage(lod_want[c], W, cutoff)
            want_segs = ol.oracle_roh_segments(cov, fp[1], *cen, W, MG, frac)
            for k, i in enumerate(sample):
                got = [(int(a), int(b)) for _, _, a, b in segs[(segs[:, 0] == i) & (segs[:, 1] == c)]]
                assert got == [(a, b) for kk, a, b in want_segs if kk == k], ("roh segments", c, i)
                n_checked += len(got)
        assert n_checked > 20
        del segs
        lod_want.clear()

        # ---- LD weights from a 500-individual subsample (--ld-subsample 500), on the device
        sub = np.sort(np.random.default_rng(9).choice(nind, size=500, replace=False)).astype(np.int32)
        ld = panel.compute_ld(W, sub_idx=sub)                              # 8 GB back to the host
        assert ld.shape == (nloci, W)
        for (glo, n), gsl in zip(ld_slices, ld_geno):
            want_ld = ol.oracle_hr2_ld(gsl, W, sub)
            assert ol.bits_equal(ld[glo:glo + n - W + 1], want_ld[:n - W + 1]), ("ld", glo)

        # ---- wLOD from those weights
        def want_wlod(c, lo, hi, fp, cen):
            return ol.oracle_calc_wlod(np.ascontiguousarray(geno_s[lo:hi][:, wsel]), *fp, spec.gpos[lo:hi], ld[lo:hi], *cen,
                                       W, ERR, MG, 1e-9, 7, threads=16)

        run(lambda: panel.wlod_windows_device(out.data_ptr(), W, ERR, MG, 7, 1e-9), "wlod", want_wlod, sel=wsel)

        # ---- TGLS (GQ dictionary codes -> term matrix [blk][SNP][64], 5 GB per block) and GL-weighted wLOD
        panel.release_scratch()

        def want_tgls(c, lo, hi, fp, cen):
            return ol.oracle_calc_lod(np.ascontiguousarray(geno_s[lo:hi]), *fp, *cen, W, ERR, MG,
                                      gl=np.ascontiguousarray(gl_s[lo:hi]), threads=16)

        def want_wlod_gl(c, lo, hi, fp, cen):
            return ol.oracle_calc_wlod(np.ascontiguousarray(geno_s[lo:hi][:, wsel]), *fp, spec.gpos[lo:hi], ld[lo:hi], *cen,
                                       W, ERR, MG, 1e-9, 7, gl=np.ascontiguousarray(gl_s[lo:hi][:, wsel]), threads=16)

        run(lambda: panel.lod_windows_device(out.data_ptr(), W, ERR, MG, use_gl=True), "tgls", want_tgls)
        assert panel.tgls_mode()[0] == 1
        run(lambda: panel.wlod_windows_device(out.data_ptr(), W, ERR, MG, 7, 1e-9, use_gl=True), "wlod gl", want_wlod_gl, sel=wsel)
        st = panel.stats()
        assert st["n_stall_reruns"] == 0 and st["n_count_timeouts"] == 0      # the strip kernel's waits all ended in time
        del ld

        # ---- continuous likelihoods at this size: 100 GB of values, converted to terms in place
        gen = torch.Generator(device=dev)
        gen.manual_seed(77)
        for l0 in range(0, nloci, 65536):
            rows = min(65536, nloci - l0)
            x = -0.3 * torch.rand((rows, nind), generator=gen, device=dev, dtype=torch.float64)     # GL: log10 likelihood
            gl = 1.0 - torch.pow(torch.tensor(10.0, dtype=torch.float64, device=dev), x)
            gl = torch.where(gl <= 0, torch.full_like(gl, 1e-16), gl)
            torch.cuda.synchronize()
            panel.set_gl_device(gl.data_ptr(), gl.shape[1], l0, rows)
            gl_s[l0:l0 + rows] = gl[:, sample].cpu().numpy()
            del x, gl
        assert panel.tgls_mode()[0] == 2
        run(lambda: panel.lod_windows_device(out.data_ptr(), W, ERR, MG, use_gl=True), "tgls continuous", want_tgls,
            whole=False)
        assert panel.tgls_mode() == (2, 1)
    finally:
        panel.close()
        del out
        torch.cuda.empty_cache()


def test_5M_by_5k_four_window_sizes(gpu_ctx):
    """config 3: --winsize-multi 50 100 200 300 on the resident panel, full scores (200 GB per size) and the
    thinned feed of each size for a --kde-subsample-like draw"""
    import torch
    nloci, nind = 5_000_000, 5000
    spec = synth.PanelSpec(nloci, seed=20260103, max_gap=MG)
    dev = torch.device("cuda", 0)
    sample = [0, 63, 64, 1342, 1344, 2559, 4991, 4992, 4999]     # row 1343 of chromosome 1's block starts beyond 4 GB
    panel, geno_s, _ = load_panel(gpu_ctx, spec, nind, sample)
    out = buf = None
    try:
        base, pitch, total = panel.out_layout(32, nind)
        assert max(int(p) for p in pitch) * 8 * 1344 > (1 << 32)
        buf = gpu_ctx.alloc_scores(total)          # 200 GB from the library's allocator (garlic_device_alloc)
        out = buf.tensor()
        for W in (50, 100, 200, 300):
            out.fill_(float("nan"))
            torch.cuda.synchronize()
            panel.lod_windows_device(out.data_ptr(), W, ERR, MG)
            torch.cuda.synchronize()
            if W == 300:
                check_whole_output(spec, out, base, pitch, nind, W, f"W={W}")
            wants = []
            for c in range(spec.nchr):
                lo, hi, fp, cen = chrom_args(spec, c)
                want = ol.oracle_calc_lod(np.ascontiguousarray(geno_s[lo:hi]), *fp, *cen, W, ERR, MG, threads=16)
                assert ol.bits_equal(sampled_rows(out, base, pitch, nind, c, hi - lo, sample), want), (W, c)
                wants.append(want)
            feed, per_chr = panel.lod_feed(W, ERR, MG, W, ind_idx=np.array(sample))
            want = [ol.oracle_flatten_subset(w, W, np.arange(len(sample))) for w in wants]
            assert [len(w) for w in want] == list(per_chr), W
            assert ol.bits_equal(feed, np.concatenate(want)), W
    finally:
        panel.close()
        del out
        if buf is not None:
            buf.free()
        torch.cuda.empty_cache()
