"""GPU: garlic_device_alloc / garlic_device_free (score matrices with reproducible placement, include/garlic_hip.h):
a buffer from there takes the scores like any device pointer, can be read back, freed, allocated again; the
plain-hipMalloc fallback behaves the same."""
import numpy as np
import pytest

from garlic_amd import abi
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("plain", [False, True])
def test_scores_into_a_library_buffer(gpu_ctx, plain, monkeypatch):
    import torch
    if plain:
        monkeypatch.setenv("GARLIC_ALLOC_PLAIN", "1")
    rng = np.random.default_rng(3)
    W, mg, sizes, nind = 20, 200000, [900, 333], 70
    chroms = [ol.random_panel(rng, n, nind, max_gap=mg) for n in sizes]
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms])
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        base, pitch, total = panel.out_layout(32, nind)
        for _ in range(2):                      # allocate, use, free -- twice
            buf = gpu_ctx.alloc_scores(total)
            assert buf.ptr and buf.ptr % 256 == 0
            panel.lod_windows_device(buf.ptr, W, 0.001, mg, pitch_align=32)
            gpu_ctx.synchronize()
            host = buf.tensor().cpu().numpy()
            for c, (g, f, p, cs, ce) in enumerate(chroms):
                got = host[base[c]: base[c] + nind * pitch[c]].reshape(nind, pitch[c])[:, :sizes[c]]
                assert ol.bits_equal(np.ascontiguousarray(got), ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, mg)), c
            buf.free()
            assert buf.ptr is None
            buf.free()                          # idempotent


def test_alloc_argument_checks(gpu_ctx):
    import ctypes as C
    p = C.c_void_p()
    assert abi.lib().garlic_device_alloc(gpu_ctx.handle, 0, C.byref(p)) != abi.OK
    assert abi.lib().garlic_device_alloc(None, 1024, C.byref(p)) != abi.OK
    assert abi.lib().garlic_device_free(gpu_ctx.handle, None) == abi.OK


def test_alloc_free_cycles_reuse_mapped_buffers(gpu_ctx):
    """20 allocate / use / free cycles of two sizes: a freed buffer stays mapped in the pool and is handed out again (no
    remap -- a virtual range given new physical memory lost part of the first kernel's writes on ROCm 7.2), so the
    reserved address space stops growing after the first cycle, and the FIRST kernel into a reused buffer writes every
    element: the whole matrix is compared with the oracle each time."""
    rng = np.random.default_rng(5)
    W, mg, sizes, nind = 20, 200000, [3000, 500], 200
    chroms = [ol.random_panel(rng, n, nind, max_gap=mg) for n in sizes]
    want = [ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, mg) for g, f, p, cs, ce in chroms]
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms])
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        base, pitch, total = panel.out_layout(32, nind)
        reserved, seen = [], set()
        for cycle in range(20):
            extra = gpu_ctx.alloc_scores(3 * total) if cycle % 3 == 0 else None      # a second size in the pool
            buf = gpu_ctx.alloc_scores(total)
            seen.add(buf.ptr)
            panel.lod_windows_device(buf.ptr, W, 0.001, mg, pitch_align=32)          # the first kernel after (re)use
            gpu_ctx.synchronize()
            host = buf.tensor().cpu().numpy()
            for c in range(len(sizes)):
                got = host[base[c]: base[c] + nind * pitch[c]].reshape(nind, pitch[c])[:, :sizes[c]]
                assert ol.bits_equal(np.ascontiguousarray(got), want[c]), (cycle, c)
            buf.tensor().fill_(float("nan"))     # whoever gets the buffer next must write everything again
            gpu_ctx.synchronize()
            buf.free()
            if extra is not None:
                extra.free()
            live, pooled, res = gpu_ctx.alloc_stats()
            reserved.append(res)
        assert len(seen) <= 2, seen                       # the same mapped ranges come back
        assert reserved[-1] == reserved[3], reserved      # no growth once both sizes are in the pool


def test_library_owned_scratch_lands_in_the_fast_placement(gpu_ctx):
    """C2 shape: the host-output call's own score scratch is chosen among candidates by timing the real kernel
    (garlic_panel_alloc_scores at first use), so a caller that hands over host buffers runs the kernel within 8 % of
    the best candidate an explicit garlic_panel_alloc_scores sees."""
    import torch
    from garlic_amd import synth
    nloci, nind, W, mg = 1_000_000, 1000, 100, 200000
    spec = synth.PanelSpec(nloci, seed=20260102, max_gap=mg)
    dev = torch.device("cuda", 0)
    with abi.Panel(gpu_ctx, spec.chr_nloci, nind) as panel:
        panel.set_map(spec.pos, spec.centro_start, spec.centro_end)
        panel.set_freq(spec.freq)
        for l0, g in synth.genotype_chunks(spec, nind, dev):
            torch.cuda.synchronize()
            panel.set_genotypes_device(g.data_ptr(), g.shape[1], l0, g.shape[0])
        buf, cand_ms = panel.alloc_scores(W, 0.001, mg, candidates=6)
        assert len(cand_ms) == 6 and min(cand_ms) > 0
        panel.lod_windows(W, 0.001, mg)          # first host-output call (8 GB of scores): its scratch is chosen now
        ms = []
        for _ in range(2):
            panel.lod_windows(W, 0.001, mg)
            ms.append(panel.stats()["chain_kernel_ms"])
        buf.free()
        assert min(ms) <= 1.08 * min(cand_ms), (ms, cand_ms)


def test_alloc_scores_takes_further_rounds_until_the_fast_placement_or_the_budget(gpu_ctx, monkeypatch):
    """A round whose best candidate does not take its score bytes at 0.74 of the HBM peak (a panel this small never
    does) is followed by another from fresh memory, three at most; one buffer is kept, the others wait in the pool, the
    scores written into the kept one are the oracle's, and garlic_panel_alloc_scores_info says what was drawn."""
    rng = np.random.default_rng(11)
    W, mg, sizes, nind = 20, 200000, [1500, 700], 70
    chroms = [ol.random_panel(rng, n, nind, max_gap=mg) for n in sizes]
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms])
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        base, pitch, total = panel.out_layout(32, nind)
        gpu_ctx.trim()
        live0, pooled0, _ = gpu_ctx.alloc_stats()
        buf, ms3 = panel.alloc_scores(W, 0.001, mg, candidates=3)
        live1, pooled1, _ = gpu_ctx.alloc_stats()
        assert len(ms3) == 3 and min(ms3) > 0
        info = panel.alloc_scores_info()
        assert info["candidates_drawn"] == 9 and info["rounds"] == 3 and not info["reached_target"]
        assert info["best_ms"] <= min(ms3) <= info["worst_ms"] and info["best_ms"] <= info["median_ms"] <= info["worst_ms"]
        assert live1 - live0 >= total * 8 and live1 - live0 < 2 * total * 8 + (4 << 20)      # one buffer kept ...
        assert pooled1 - pooled0 >= 8 * total * 8                  # ... eight candidates of three rounds in the pool
        monkeypatch.setenv("GARLIC_ALLOC_ROUNDS", "1")
        buf1, ms1 = panel.alloc_scores(W, 0.001, mg, candidates=2)
        assert len(ms1) == 2 and min(ms1) > 0
        assert panel.alloc_scores_info()["candidates_drawn"] == 2
        for b in (buf, buf1):
            panel.lod_windows_device(b.ptr, W, 0.001, mg, pitch_align=32)
            gpu_ctx.synchronize()
            host = b.tensor().cpu().numpy()
            for c, (g, f, p, cs, ce) in enumerate(chroms):
                got = host[base[c]: base[c] + nind * pitch[c]].reshape(nind, pitch[c])[:, :sizes[c]]
                assert ol.bits_equal(np.ascontiguousarray(got), ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, mg)), c
            b.free()
