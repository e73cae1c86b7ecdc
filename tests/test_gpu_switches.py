"""GPU: every environment switch of libgarlic_hip.so that selects another kernel or path for the same result -- the
forms earlier rounds shipped and later ones replaced, the fall-backs a shape can reach, the debugging knobs -- through a
slice of tools/soak.py (random panels, every variant: full scores, feeds, LD weights, wLOD, likelihoods, coverage counts
and ROH segments, each against the CPU oracle bit for bit).  A switch without a reader here or in another test does not
exist in the library (tests/test_abi.py::test_every_switch_has_a_test)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

SWITCHES = [
    "GARLIC_FEED_NO_ASM",               # thinned feed / coverage bits: every tile through the compiler-generated path
    "GARLIC_FEED_PER_CU=1",             # ... one persistent workgroup per CU
    "GARLIC_COVERAGE_UNFUSED",          # coverage counts from a score matrix (the form every shape can fall back to)
    "GARLIC_TGLS_NO_RING",              # likelihoods: round 1's two-wave chain instead of the ring kernel
    "GARLIC_TGLS_CONTINUOUS",           # ... 8-byte values although a dictionary would do
    "GARLIC_GL_TERMS_GATHER",           # ... term matrix by gather
    "GARLIC_GL_NO_TERMS",               # ... no term matrix (terms looked up in the chain)
    "GARLIC_WLOD_GENERIC",              # wLOD: the generic kernel
    "GARLIC_WLOD_ONE_BLOCK",            # ... one block per wave
    "GARLIC_WLOD_NO_PF",                # ... no weight touches
    "GARLIC_WLOD_NO_PATCH",             # ... stores without the write-out patch
    "GARLIC_WLOD_SMALL_TILES",          # ... narrow windows: round 2's tile kernel
    "GARLIC_WLOD_SMALL_GENERIC",        # ... narrow windows: the generic kernel
    "GARLIC_WLOD_GL_NO_RING",           # weighted with likelihoods: no LDS rings
    "GARLIC_WLOD_GL_NO_PATCH",
    "GARLIC_WLOD_STRIP_NARROW_ONLY",    # ... strips only up to W = 113
    "GARLIC_WLOD_STRIP_TWO_PER_CU",     # ... the 96-VGPR strip kernel (two workgroups per CU) where the 80-VGPR one would run
    "GARLIC_LD_UNFUSED",                # LD weights: pair table, then hr2 table (the two steps a sharded panel takes)
    "GARLIC_LD_PAIR_NO_MFMA",           # ... pair counts without the matrix cores
    "GARLIC_LD_PAIR_TILED",
    "GARLIC_LD_PAIR_FLAT",
    "GARLIC_LD_PAIR_L2",
    "GARLIC_LD_LANE_STAGE",
    "GARLIC_LD_NO_PLANE_CACHE",
    "GARLIC_LD_NO_FLAT",
    "GARLIC_LD_HR2_PLAIN",
    "GARLIC_NO_PLACEMENT",              # score scratch without the placement probe
]


@pytest.mark.parametrize("switch", SWITCHES)
def test_soak_slice_under_switch(gpu_ctx, switch, monkeypatch, capsys):
    import soak
    name, _, value = switch.partition("=")
    monkeypatch.setenv(name, value or "1")
    checks, fails = soak.soak(gpu_ctx, 3, 20260400 + sum(map(ord, name)), verbose=False)
    out = capsys.readouterr().out
    assert fails == 0, out[-3000:]
    assert checks > 30
