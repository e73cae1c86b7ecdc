"""GPU: a 60-panel slice of tools/soak.py in the driver-run suite -- random shapes around the wave, tile and window
sizes (chromosomes of 1, W-1, W, W+1 SNPs, 1..260 individuals, windows 2..130) through EVERY variant: full scores in
both layouts, thinned and subset feeds, the multi-size feed, hr2 and phased r2 LD weights with subsamples, wLOD, TGLS
(dictionary and continuous likelihoods), GL-weighted wLOD with short strips -- each against the CPU oracle bit for bit."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("seed", [20260301, 20260302])
def test_soak_slice(gpu_ctx, seed, capsys):
    import soak
    checks, fails = soak.soak(gpu_ctx, 30, seed, verbose=False)
    out = capsys.readouterr().out
    assert fails == 0, out[-3000:]
    assert checks > 300
