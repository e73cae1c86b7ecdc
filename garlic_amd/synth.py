"""Deterministic synthetic SNP x individual panels for bench.py and the
full-size property tests.  Not part of the hot path.

Per-SNP data (positions, allele frequencies, genetic map) are made on the host with numpy;
genotypes are drawn chunk by chunk with torch on whatever device is asked for (the GPU for the
bench, so a 1M x 1k panel never exists as host memory).
"""
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

HG19_LEN = [249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663,
            146364022, 141213431, 135534747, 135006516, 133851895, 115169878, 107349540,
            102531392, 90354753, 81195210, 78077248, 59128983, 63025520, 48129895, 51304566]


def hg19_centromeres():
    with open(os.path.join(_HERE, "centromeres.json")) as f:
        t = json.load(f)["hg19"]
    return [tuple(t["chr%d" % (c + 1)]) for c in range(22)]


class PanelSpec:
    """Per-SNP description of a synthetic panel: 22 autosomes, SNP counts proportional to the
    hg19 chromosome lengths, exponential spacing, no SNP inside a centromere, one injected
    > max_gap hole per chromosome, allele frequency U(0.05, 0.95), 1 cM/Mb +-20 % genetic map."""

    def __init__(self, nloci, seed, max_gap=200000, nchr=22):
        rng = np.random.default_rng(seed)
        lens = np.array(HG19_LEN[:nchr], dtype=np.float64)
        counts = np.maximum(1, np.floor(nloci * lens / lens.sum()).astype(np.int64))
        counts[0] += nloci - counts.sum()
        assert counts.sum() == nloci and counts.min() >= 1
        cen = hg19_centromeres()[:nchr]
        pos, gpos = [], []
        for c in range(nchr):
            n = int(counts[c])
            cs, ce = cen[c]
            usable = HG19_LEN[c] - (ce - cs + 1) - max_gap - 2000
            gaps = rng.exponential(1.0, size=n)
            gaps *= 0.97 * usable / gaps.sum()
            p = np.cumsum(np.maximum(1, gaps).astype(np.int64))
            if n > 4:  # one hole wider than max_gap
                k = int(rng.integers(n // 4, 3 * n // 4))
                p[k:] += max_gap + 1000
            p = np.where(p >= cs, p + (ce - cs + 1), p)  # nothing inside the centromere
            assert (np.diff(p) > 0).all() and p[-1] < 2**31
            pos.append(p.astype(np.int32))
            rate = 1e-6 * rng.uniform(0.8, 1.2, size=n)  # cM per bp
            gpos.append(np.cumsum(np.diff(p, prepend=0) * rate))
        self.nchr = nchr
        self.chr_nloci = counts.astype(np.int32)
        self.chr_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.pos = np.concatenate(pos)
        self.gpos = np.concatenate(gpos)
        self.centro_start = np.array([c[0] for c in cen], dtype=np.int32)
        self.centro_end = np.array([c[1] for c in cen], dtype=np.int32)
        self.freq = rng.uniform(0.05, 0.95, size=nloci)
        self.nloci = int(nloci)
        self.seed = int(seed)
        self.max_gap = int(max_gap)


def genotype_chunks(spec, nind, device, chunk=65536, ind_offset=0, miss=0.01, tracts=8):
    """Yields (locus_begin, int16 tensor [rows][nind]) : HWE draws from spec.freq with `miss`
    missing (-9) and `tracts` planted homozygous tracts per individual (hets forced to the
    nearer homozygote) so the LOD distribution is bimodal.  ind_offset makes shards of one big
    panel distinct (rank r draws individuals [r*nind, (r+1)*nind))."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(spec.seed * 1000003 + ind_offset)
    freq = torch.from_numpy(spec.freq).to(device)
    # tract [start, start+len) in global SNP coordinates, per individual
    t_start = torch.randint(0, max(1, spec.nloci - 1), (tracts, nind), generator=g, device=device)
    t_len = torch.randint(500, 3000, (tracts, nind), generator=g, device=device)
    for l0 in range(0, spec.nloci, chunk):
        l1 = min(spec.nloci, l0 + chunk)
        p = freq[l0:l1, None]
        u = torch.rand((l1 - l0, nind), generator=g, device=device, dtype=torch.float32)
        q0 = ((1 - p) * (1 - p)).float()
        q1 = (q0 + 2 * p * (1 - p)).float()
        geno = (u >= q0).to(torch.int16) + (u >= q1).to(torch.int16)
        loc = torch.arange(l0, l1, device=device)[:, None]
        in_tract = torch.zeros((l1 - l0, nind), dtype=torch.bool, device=device)
        for k in range(tracts):
            in_tract |= (loc >= t_start[k][None, :]) & (loc < (t_start[k] + t_len[k])[None, :])
        hom = torch.where(p > 0.5, 2, 0).to(torch.int16).expand(-1, nind)
        geno = torch.where(in_tract & (geno == 1), hom, geno)
        m = torch.rand((l1 - l0, nind), generator=g, device=device, dtype=torch.float32) < miss
        geno = torch.where(m, torch.full_like(geno, -9), geno)
        yield l0, geno.contiguous()
