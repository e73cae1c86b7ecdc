"""The reference-side binding of INTEGRATION.md (oracle/ref_binding.cpp) is real code:

* CPU (build container, where /root/reference exists): it compiles against the reference's own garlic-roh.h /
  garlic-data.h / garlic-centromeres.h and links, -z defs, with the reference's own objects and libgarlic_hip.so;
  INTEGRATION.md's code block is that file's marked region.
* GPU: the same vector<HapData*>* .. centromere* go to the reference's calcLODWindows / calcwLODWindows and to the
  binding; the WinData rows are memcmp'd and both results freed with the reference's releaseWinData."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libgarlic_ref_hip.so")


def test_integration_md_is_generated_from_the_binding(tmp_path):
    env = {k: v for k, v in os.environ.items() if not k.startswith("GARLIC_")}
    env["GARLIC_GEN_OUT"] = str(tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_integration.py")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(tmp_path / "INTEGRATION.md").read() == open(os.path.join(ROOT, "INTEGRATION.md")).read(), \
        "INTEGRATION.md is stale: run tools/gen_integration.py"


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference sources only exist in the build container")
def test_binding_compiles_against_the_reference_headers_and_links(tmp_path):
    out = tmp_path / "ref"
    r = subprocess.run(["make", "-s", "-f", os.path.join(ROOT, "oracle", "Makefile"), f"OUT={out}", "ref"],
                       capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    so = out / "libgarlic_ref_hip.so"
    assert so.exists()
    syms = subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True).stdout
    assert "refbind_compare_lod" in syms and "refbind_compare_wlod" in syms and "refbind_compare_roh" in syms
    undefined = subprocess.run(["nm", "-D", "--undefined-only", str(so)], capture_output=True, text=True).stdout
    assert "garlic_lod_windows" in undefined and "garlic_wlod_windows" in undefined   # from libgarlic_hip.so, nothing stubbed
    assert "garlic_roh_segments" in undefined


def _lib():
    lib = C.CDLL(SO)
    i32p, f64p, i16p = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int16)
    lib.refbind_compare_lod.restype = C.c_long
    lib.refbind_compare_lod.argtypes = [C.c_int, i32p, C.c_int, i16p, f64p, i32p, i32p, i32p, f64p, C.c_int, C.c_double, C.c_int]
    lib.refbind_compare_wlod.restype = C.c_long
    lib.refbind_compare_wlod.argtypes = [C.c_int, i32p, C.c_int, i16p, f64p, i32p, f64p, i32p, i32p, f64p, C.c_int, C.c_double,
                                         C.c_int, C.c_int, C.c_double, C.c_int]
    lib.refbind_compare_roh.restype = C.c_long
    lib.refbind_compare_roh.argtypes = [C.c_int, i32p, C.c_int, i16p, f64p, i32p, f64p, i32p, i32p, f64p, C.c_int, C.c_double,
                                        C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_long)]
    return lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(SO), reason="oracle/_ref/libgarlic_ref_hip.so is built in the build container (make -C oracle ref)")
def test_binding_equals_the_reference_on_the_references_own_structs(gpu_ctx):      # (the fixture: torch finds the device first, conftest.py)
    rng = np.random.default_rng(404)
    lib = _lib()
    mg = 200000
    for sizes, nind, W in (([700, 90, 33], 45, 30), ([400, 260], 130, 60), ([150], 3, 2)):
        chroms = [ol.random_panel(rng, n, nind, max_gap=mg, gaps=2 if n > 300 else 0) for n in sizes]
        nl = np.array(sizes, dtype=np.int32)
        geno = np.ascontiguousarray(np.concatenate([c[0] for c in chroms], axis=0), dtype=np.int16)
        freq = np.ascontiguousarray(np.concatenate([c[1] for c in chroms]))
        pos = np.ascontiguousarray(np.concatenate([c[2] for c in chroms]), dtype=np.int32)
        gpos = np.ascontiguousarray(pos.astype(np.float64) * 1e-6)
        cs = np.array([c[3] for c in chroms], dtype=np.int32)
        ce = np.array([c[4] for c in chroms], dtype=np.int32)
        cs[-1] = -1                                            # last chromosome unknown to the centromere table: (0, 0)
        gl = np.ascontiguousarray(rng.choice([1e-3, 0.01, 0.2, 10 ** -3.7], size=geno.shape))
        for use_gl in (None, gl):
            bad = lib.refbind_compare_lod(len(sizes), _p(nl, C.c_int32), nind, _p(geno, C.c_int16), _p(freq, C.c_double),
                                          _p(pos, C.c_int32), _p(cs, C.c_int32), _p(ce, C.c_int32), _p(use_gl, C.c_double), W,
                                          0.001, mg)
            assert bad == 0, ("calcLODWindows", sizes, W, use_gl is not None, bad)
            bad = lib.refbind_compare_wlod(len(sizes), _p(nl, C.c_int32), nind, _p(geno, C.c_int16), _p(freq, C.c_double),
                                           _p(pos, C.c_int32), _p(gpos, C.c_double), _p(cs, C.c_int32), _p(ce, C.c_int32),
                                           _p(use_gl, C.c_double), W, 0.001, mg, 7, 1e-9, 3)
            assert bad == 0, ("calcwLODWindows", sizes, W, use_gl is not None, bad)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(SO), reason="oracle/_ref/libgarlic_ref_hip.so is built in the build container (make -C oracle ref)")
def test_roh_binding_equals_the_references_calcLODWindows_plus_assembleROHWindows(gpu_ctx):
    """The reference's calcLODWindows -> assembleROHWindows and the binding's one call (garlic_roh_segments) on the same
    vector<HapData*>* .. IndData*: every individual's chr / start / stop / length lists and the pooled ROHLength, in bp
    and in cM, with and without likelihoods, thresholds from one SNP to the whole window"""
    rng = np.random.default_rng(808)
    lib = _lib()
    mg = 200000
    total = 0
    for sizes, nind, W in (([900, 90, 33], 45, 30), ([500, 260], 70, 12), ([150], 3, 2)):
        chroms = [ol.random_panel(rng, n, nind, max_gap=mg, gaps=2 if n > 300 else 0) for n in sizes]
        nl = np.array(sizes, dtype=np.int32)
        geno = np.ascontiguousarray(np.concatenate([c[0] for c in chroms], axis=0), dtype=np.int16)
        freq = np.ascontiguousarray(np.concatenate([c[1] for c in chroms]))
        pos = np.ascontiguousarray(np.concatenate([c[2] for c in chroms]), dtype=np.int32)
        gpos = np.ascontiguousarray(pos.astype(np.float64) * 1.3e-6)
        cs = np.array([c[3] for c in chroms], dtype=np.int32)
        ce = np.array([c[4] for c in chroms], dtype=np.int32)
        cs[-1] = -1
        gl = np.ascontiguousarray(rng.choice([1e-3, 0.01, 0.2, 10 ** -3.7], size=geno.shape))
        for use_gl in (None, gl):
            for cutoff, frac, cm in ((0.0, 0.25, 0), (-1.0, 1e-9, 1), (0.5, 1.0, 0), (-3.0, 0.6, 1)):
                n = C.c_long(0)
                bad = lib.refbind_compare_roh(len(sizes), _p(nl, C.c_int32), nind, _p(geno, C.c_int16), _p(freq, C.c_double),
                                              _p(pos, C.c_int32), _p(gpos, C.c_double), _p(cs, C.c_int32), _p(ce, C.c_int32),
                                              _p(use_gl, C.c_double), W, 0.001, mg, cutoff, frac, cm, C.byref(n))
                assert bad == 0, ("assembleROHWindows", sizes, W, use_gl is not None, cutoff, frac, cm, bad)
                total += n.value
    assert total > 300
