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
