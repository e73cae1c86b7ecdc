"""GPU: the strip form of the GL-weighted wLOD kernel (wlod_strip_kernel.hpp: 7 compute waves + a loader per
workgroup for W <= 113, 15 + 1 for W <= 241, the two blocks' term rows through shared LDS rings) against the oracle (garlic-roh.cpp:204-277 with
USE_GL) and against the tile kernel it replaces -- strips shorter than the workgroup has waves, many strips per
chromosome, a last pair with one block, block-aligned sub-ranges, windows up to the widest the strip form takes."""
import os

import numpy as np
import pytest

from garlic_amd import abi
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _panel(rng, sizes, nind, W, mg):
    chroms = [ol.random_panel(rng, n, nind, max_gap=mg, gaps=3 if n > 300 else 0) for n in sizes]
    gpos = [np.cumsum(np.diff(c[2], prepend=0) * 1e-6 * rng.uniform(0.8, 1.2, size=c[2].shape[0])) for c in chroms]
    lds = [rng.uniform(1.0, max(2.0, W / 4.0), size=(n, W)) for n in sizes]
    err = [rng.choice([1e-16, 1e-3, 0.01, 0.2, 1.0], size=c[0].shape) for c in chroms]
    return chroms, gpos, lds, err


@pytest.mark.parametrize("W,groups", [(16, None), (33, 1), (100, 5), (100, None), (113, 9), (64, 2), (114, 3), (200, None),
                                      (241, 20), (242, None)])
def test_strip_kernel_against_oracle_and_tile_kernel(gpu_ctx, W, groups, monkeypatch):
    rng = np.random.default_rng(4100 + W + (groups or 0))
    mg = 60000
    sizes = [1500, 40, W, W + 1, 700, 1]
    nind = 200 if W != 100 else 130            # 4 blocks = 2 pairs; 3 blocks: the second pair has one block
    chroms, gpos, lds, err = _panel(rng, sizes, nind, W, mg)
    if groups is not None:
        monkeypatch.setenv("GARLIC_WLOD_STRIP_GROUPS", str(groups))
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms],
                      gpos=np.concatenate(gpos))
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        panel.set_ld(W, np.concatenate(lds, axis=0))
        panel.set_gl(np.concatenate(err, axis=0))
        want = [ol.oracle_calc_wlod(g, f, p, gpos[c], lds[c], cs, ce, W, 0.001, mg, 1e-9, 7, gl=err[c])
                for c, (g, f, p, cs, ce) in enumerate(chroms)]
        for pa, i0, cnt in ((32, 0, nind), (32, 64, nind - 64), (2, 64, 60), (1, 0, nind)):
            out = panel.wlod_windows(W, 0.001, mg, 7, 1e-9, pitch_align=pa, ind_begin=i0, ind_count=cnt, use_gl=True)
            for c in range(len(sizes)):
                assert ol.bits_equal(np.ascontiguousarray(out[c]), want[c][i0:i0 + cnt]), (pa, i0, c)
        assert panel.stats()["n_stall_reruns"] == 0        # no strip wave ran out of its poll budget
        monkeypatch.setenv("GARLIC_WLOD_GL_NO_STRIP", "1")
        out = panel.wlod_windows(W, 0.001, mg, 7, 1e-9, pitch_align=32, use_gl=True)
        for c in range(len(sizes)):
            assert ol.bits_equal(np.ascontiguousarray(out[c]), want[c]), ("tile kernel", c)


@pytest.mark.parametrize("W", [40, 100, 113])
def test_both_narrow_strip_kernels_agree_with_the_oracle(gpu_ctx, W, monkeypatch):
    """W <= 113, scores into 16-B aligned rows: the 80-VGPR kernel (three workgroups per CU, write-out inside the generated
    loop) by default, the 96-VGPR one under GARLIC_WLOD_STRIP_TWO_PER_CU -- odd and even chromosome lengths (the last
    window alone in its 16-B piece), a last block of 2 individuals, windows without a score between scored ones"""
    rng = np.random.default_rng(5200 + W)
    mg = 30000
    sizes = [2 * W + 17, 2 * W + 18, 1203, W]
    nind = 130
    chroms, gpos, lds, err = _panel(rng, sizes, nind, W, mg)
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms],
                      gpos=np.concatenate(gpos))
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        panel.set_ld(W, np.concatenate(lds, axis=0))
        panel.set_gl(np.concatenate(err, axis=0))
        want = [ol.oracle_calc_wlod(g, f, p, gpos[c], lds[c], cs, ce, W, 0.001, mg, 1e-9, 7, gl=err[c])
                for c, (g, f, p, cs, ce) in enumerate(chroms)]
        for two in (False, True):
            if two:
                monkeypatch.setenv("GARLIC_WLOD_STRIP_TWO_PER_CU", "1")
            for pa in (32, 2):
                out = panel.wlod_windows(W, 0.001, mg, 7, 1e-9, pitch_align=pa, use_gl=True)
                for c in range(len(sizes)):
                    assert ol.bits_equal(np.ascontiguousarray(out[c]), want[c]), (two, pa, c)
        assert panel.stats()["n_stall_reruns"] == 0


def test_strip_kernel_repeated_calls_and_plan_reuse(gpu_ctx, monkeypatch):
    """the same call twice (plan reused), then with the strip form switched off and on again (plan rebuilt)"""
    rng = np.random.default_rng(77)
    W, mg, sizes, nind = 80, 10 ** 9, [3000, 900], 128
    chroms, gpos, lds, err = _panel(rng, sizes, nind, W, mg)
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms],
                      gpos=np.concatenate(gpos))
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        panel.set_ld(W, np.concatenate(lds, axis=0))
        panel.set_gl(np.concatenate(err, axis=0))
        want = [ol.oracle_calc_wlod(g, f, p, gpos[c], lds[c], cs, ce, W, 0.001, mg, 1e-9, 7, gl=err[c])
                for c, (g, f, p, cs, ce) in enumerate(chroms)]
        for step in range(4):
            if step == 2:
                monkeypatch.setenv("GARLIC_WLOD_GL_NO_STRIP", "1")
            if step == 3:
                monkeypatch.delenv("GARLIC_WLOD_GL_NO_STRIP")
            out = panel.wlod_windows(W, 0.001, mg, 7, 1e-9, pitch_align=32, use_gl=True)
            for c in range(len(sizes)):
                assert ol.bits_equal(np.ascontiguousarray(out[c]), want[c]), (step, c)


def test_a_stalled_strip_launch_is_repaired_on_the_device_and_counted(gpu_ctx, monkeypatch):
    """GARLIC_WLOD_STRIP_FORCE_RERUN sets the strip kernel's stall flag after every launch: the tile form enqueued behind
    it (it runs only when the flag is set; no copy back, no synchronisation) recomputes the scores, and
    garlic_call_stats.n_stall_reruns says how often that happened -- 0 without the switch"""
    rng = np.random.default_rng(78)
    W, mg, sizes, nind = 80, 10 ** 9, [2500, 700], 128
    chroms, gpos, lds, err = _panel(rng, sizes, nind, W, mg)
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms],
                      gpos=np.concatenate(gpos))
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        panel.set_ld(W, np.concatenate(lds, axis=0))
        panel.set_gl(np.concatenate(err, axis=0))
        want = [ol.oracle_calc_wlod(g, f, p, gpos[c], lds[c], cs, ce, W, 0.001, mg, 1e-9, 7, gl=err[c])
                for c, (g, f, p, cs, ce) in enumerate(chroms)]
        out = panel.wlod_windows(W, 0.001, mg, 7, 1e-9, pitch_align=32, use_gl=True)
        assert panel.stats()["n_stall_reruns"] == 0
        monkeypatch.setenv("GARLIC_WLOD_STRIP_FORCE_RERUN", "1")
        for k in (1, 2):
            out = panel.wlod_windows(W, 0.001, mg, 7, 1e-9, pitch_align=32, use_gl=True)
            for c in range(len(sizes)):
                assert ol.bits_equal(np.ascontiguousarray(out[c]), want[c]), (k, c)
            assert panel.stats()["n_stall_reruns"] == k
        monkeypatch.delenv("GARLIC_WLOD_STRIP_FORCE_RERUN")
        panel.wlod_windows(W, 0.001, mg, 7, 1e-9, pitch_align=32, use_gl=True)
        assert panel.stats()["n_stall_reruns"] == 2
