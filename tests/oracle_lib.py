"""ctypes bindings for the CPU oracle (oracle/liblod_oracle.so) and, when it was built in the
build container, the real reference (oracle/_ref/libgarlic_ref.so).

TEST INFRASTRUCTURE ONLY: nothing under garlic_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liblod_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libgarlic_ref.so")
MISSING = -9999.0

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_sp = C.POINTER(C.c_int16)
_bp = C.POINTER(C.c_uint8)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def build_oracle():
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
            os.path.join(ORACLE_DIR, "lod_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
    return ORACLE_SO


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(build_oracle())
        lib.oracle_lod.restype = C.c_double
        lib.oracle_lod.argtypes = [C.c_int, C.c_double, C.c_double]
        lib.oracle_in_gap.restype = C.c_int
        lib.oracle_in_gap.argtypes = [C.c_int] * 4
        lib.oracle_nomut.restype = C.c_double
        lib.oracle_nomut.argtypes = [C.c_double] * 3
        lib.oracle_norec.restype = C.c_double
        lib.oracle_norec.argtypes = [C.c_double] * 2
        lib.oracle_tgls_to_error.restype = C.c_double
        lib.oracle_tgls_to_error.argtypes = [C.c_double, C.c_int]
        lib.oracle_calc_lod.restype = None
        lib.oracle_calc_lod.argtypes = [C.c_int, C.c_int, _sp, _dp, _ip, _dp, C.c_int, C.c_int,
                                        C.c_int, C.c_double, C.c_int, _dp]
        lib.oracle_calc_lod_mt.restype = None
        lib.oracle_calc_lod_mt.argtypes = [C.c_int, C.c_int, _sp, _dp, _ip, _dp, C.c_int, C.c_int,
                                           C.c_int, C.c_double, C.c_int, C.c_int, _dp]
        lib.oracle_calc_wlod.restype = None
        lib.oracle_calc_wlod.argtypes = [C.c_int, C.c_int, _sp, _dp, _ip, _dp, _dp, _dp,
                                         C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                         C.c_double, C.c_int, C.c_int, _dp]
        lib.oracle_geno_freq.restype = None
        lib.oracle_geno_freq.argtypes = [C.c_int, C.c_int, _sp, _dp]
        lib.oracle_hr2_ld.restype = None
        lib.oracle_hr2_ld.argtypes = [C.c_int, C.c_int, _sp, _dp, C.c_int, _ip, C.c_int, _dp]
        lib.oracle_r2_ld.restype = None
        lib.oracle_r2_ld.argtypes = [C.c_int, C.c_int, _sp, _bp, _dp, C.c_int, _ip, C.c_int, _dp]
        lib.oracle_flatten.restype = C.c_int64
        lib.oracle_flatten.argtypes = [C.c_int, C.c_int, _dp, C.c_int, _dp]
        lib.oracle_flatten_subset.restype = C.c_int64
        lib.oracle_flatten_subset.argtypes = [C.c_int, C.c_int, _dp, C.c_int, _ip, C.c_int, _dp]
        lib.oracle_roh_coverage.restype = None
        lib.oracle_roh_coverage.argtypes = [C.c_int, C.c_int, _dp, C.c_int, C.c_double, _sp]
        lib.oracle_roh_segments.restype = C.c_int
        lib.oracle_roh_segments.argtypes = [C.c_int, _sp, _ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, _ip, _ip]
        lib.oracle_mask.restype = None
        lib.oracle_mask.argtypes = [C.c_int, _ip, C.c_int, C.c_int, C.c_int, C.c_int, _bp]
        _oracle = lib
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


_ref = None


def ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_SO)
        lib.ref_lod.restype = C.c_double
        lib.ref_lod.argtypes = [C.c_short, C.c_double, C.c_double]
        lib.ref_nomut.restype = C.c_double
        lib.ref_nomut.argtypes = [C.c_double] * 3
        lib.ref_norec.restype = C.c_double
        lib.ref_norec.argtypes = [C.c_double] * 2
        lib.ref_inGap.restype = C.c_int
        lib.ref_inGap.argtypes = [C.c_int] * 4
        lib.ref_calcLOD.restype = C.c_int
        lib.ref_calcLOD.argtypes = [C.c_int, C.c_int, _sp, _dp, _ip, _dp, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_double, C.c_int, _dp]
        lib.ref_calcwLOD.restype = C.c_int
        lib.ref_calcwLOD.argtypes = [C.c_int, C.c_int, _sp, _dp, _ip, _dp, _dp, _dp,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                     C.c_double, C.c_int, C.c_int, _dp]
        lib.ref_calcHR2LD.restype = C.c_int
        lib.ref_calcHR2LD.argtypes = [C.c_int, C.c_int, _sp, C.c_int, C.c_int, _ip, C.c_int, _dp, _dp]
        lib.ref_assembleROH.restype = C.c_int
        lib.ref_assembleROH.argtypes = [C.c_int, C.c_int, _dp, _ip, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                        C.c_int, C.c_double, C.c_int, _ip, _dp, _dp]
        lib.ref_calcR2LD.restype = C.c_int
        lib.ref_calcR2LD.argtypes = [C.c_int, C.c_int, _sp, _bp, _dp, C.c_int, C.c_int, _ip, C.c_int, _dp]
        lib.ref_readTGLS.restype = C.c_int
        lib.ref_readTGLS.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, _dp]
        lib.ref_flatten.restype = C.c_int
        lib.ref_flatten.argtypes = [C.c_int, C.c_int, _dp, C.c_int, _dp]
        _ref = lib
    return _ref


# ---------------------------------------------------------------- numpy-level wrappers

def _prep(geno, freq, pos, gl=None):
    geno = np.ascontiguousarray(geno, dtype=np.int16)
    freq = np.ascontiguousarray(freq, dtype=np.float64)
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    if gl is not None:
        gl = np.ascontiguousarray(gl, dtype=np.float64)
    return geno, freq, pos, gl


def oracle_calc_lod(geno, freq, pos, cS, cE, W, error, max_gap, gl=None, threads=0):
    geno, freq, pos, gl = _prep(geno, freq, pos, gl)
    nloci, nind = geno.shape
    win = np.empty((nind, nloci), dtype=np.float64)
    if threads > 0:
        oracle().oracle_calc_lod_mt(nloci, nind, _p(geno, _sp), _p(freq, _dp), _p(pos, _ip),
                                    _p(gl, _dp), cS, cE, W, error, max_gap, threads, _p(win, _dp))
    else:
        oracle().oracle_calc_lod(nloci, nind, _p(geno, _sp), _p(freq, _dp), _p(pos, _ip),
                                 _p(gl, _dp), cS, cE, W, error, max_gap, _p(win, _dp))
    return win


def oracle_calc_wlod(geno, freq, pos, gpos, ld, cS, cE, W, error, max_gap, mu, M, gl=None,
                     threads=1):
    geno, freq, pos, gl = _prep(geno, freq, pos, gl)
    gpos = np.ascontiguousarray(gpos, dtype=np.float64)
    ld = np.ascontiguousarray(ld, dtype=np.float64)
    nloci, nind = geno.shape
    win = np.empty((nind, nloci), dtype=np.float64)
    oracle().oracle_calc_wlod(nloci, nind, _p(geno, _sp), _p(freq, _dp), _p(pos, _ip),
                              _p(gpos, _dp), _p(gl, _dp), _p(ld, _dp), cS, cE, W, error, max_gap,
                              mu, M, threads, _p(win, _dp))
    return win


def oracle_mask(pos, cS, cE, W, max_gap):
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    valid = np.empty(pos.shape[0], dtype=np.uint8)
    oracle().oracle_mask(pos.shape[0], _p(pos, _ip), cS, cE, W, max_gap, _p(valid, _bp))
    return valid


def oracle_geno_freq(geno):
    geno = np.ascontiguousarray(geno, dtype=np.int16)
    hom = np.empty(geno.shape[0], dtype=np.float64)
    oracle().oracle_geno_freq(geno.shape[0], geno.shape[1], _p(geno, _sp), _p(hom, _dp))
    return hom


def oracle_hr2_ld(geno, W, idx=None):
    geno = np.ascontiguousarray(geno, dtype=np.int16)
    nloci, nind = geno.shape
    if idx is None:
        idx = np.arange(nind, dtype=np.int32)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    hom = oracle_geno_freq(geno)
    ld = np.empty((nloci, W), dtype=np.float64)
    oracle().oracle_hr2_ld(nloci, nind, _p(geno, _sp), _p(hom, _dp), W, _p(idx, _ip), idx.shape[0],
                           _p(ld, _dp))
    return ld


def oracle_r2_ld(geno, first_copy, freq, W, idx=None):
    """--phased LD weights (calcR2LD); first_copy uint8 [nloci][nind]"""
    geno = np.ascontiguousarray(geno, dtype=np.int16)
    fc = np.ascontiguousarray(first_copy, dtype=np.uint8)
    freq = np.ascontiguousarray(freq, dtype=np.float64)
    nloci, nind = geno.shape
    if idx is None:
        idx = np.arange(nind, dtype=np.int32)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    ld = np.empty((nloci, W), dtype=np.float64)
    oracle().oracle_r2_ld(nloci, nind, _p(geno, _sp), _p(fc, _bp), _p(freq, _dp), W, _p(idx, _ip),
                          idx.shape[0], _p(ld, _dp))
    return ld


def oracle_flatten(win, step):
    win = np.ascontiguousarray(win, dtype=np.float64)
    nind, nloci = win.shape
    out = np.empty(win.size, dtype=np.float64)
    n = oracle().oracle_flatten(nloci, nind, _p(win, _dp), step, _p(out, _dp))
    return out[:n].copy()


def oracle_flatten_subset(win, step, idx):
    """convertSubsetWinData2DoubleData (garlic-data.cpp:2071-2150) with the drawn individuals given"""
    win = np.ascontiguousarray(win, dtype=np.float64)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    nind, nloci = win.shape
    out = np.empty(max(1, idx.shape[0] * nloci), dtype=np.float64)
    n = oracle().oracle_flatten_subset(nloci, nind, _p(win, _dp), step, _p(idx, _ip), idx.shape[0], _p(out, _dp))
    return out[:n].copy()


def oracle_roh_coverage(win, W, cutoff):
    win = np.ascontiguousarray(win, dtype=np.float64)
    nind, nloci = win.shape
    out = np.empty((nind, nloci), dtype=np.int16)
    oracle().oracle_roh_coverage(nloci, nind, _p(win, _dp), W, cutoff, _p(out, _sp))
    return out


def oracle_roh_segments(inwin, pos, cS, cE, W, max_gap, overlap_frac):
    """second half of assembleROHWindows on coverage counts [nind][nloci] -> [(individual, start index, stop index)]"""
    inwin = np.ascontiguousarray(inwin, dtype=np.int16)
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    nind, nloci = inwin.shape
    a = np.empty(nloci + 1, dtype=np.int32)
    b = np.empty(nloci + 1, dtype=np.int32)
    out = []
    for i in range(nind):
        n = oracle().oracle_roh_segments(nloci, _p(inwin[i], _sp), _p(pos, _ip), cS, cE, W, max_gap, overlap_frac,
                                         nloci + 1, _p(a, _ip), _p(b, _ip))
        out += [(i, int(a[k]), int(b[k])) for k in range(n)]
    return out


def ref_calc_lod(geno, freq, pos, cS, cE, W, error, max_gap, gl=None, centro_known=True):
    geno, freq, pos, gl = _prep(geno, freq, pos, gl)
    nloci, nind = geno.shape
    win = np.empty((nind, nloci), dtype=np.float64)
    rc = ref().ref_calcLOD(nloci, nind, _p(geno, _sp), _p(freq, _dp), _p(pos, _ip), _p(gl, _dp),
                           cS, cE, int(centro_known), W, error, max_gap, _p(win, _dp))
    assert rc == 0
    return win


def ref_calc_wlod(geno, freq, pos, gpos, ld, cS, cE, W, error, max_gap, mu, M, gl=None,
                  threads=1, centro_known=True):
    geno, freq, pos, gl = _prep(geno, freq, pos, gl)
    gpos = np.ascontiguousarray(gpos, dtype=np.float64)
    ld = np.ascontiguousarray(ld, dtype=np.float64)
    nloci, nind = geno.shape
    win = np.empty((nind, nloci), dtype=np.float64)
    rc = ref().ref_calcwLOD(nloci, nind, _p(geno, _sp), _p(freq, _dp), _p(pos, _ip), _p(gpos, _dp),
                            _p(gl, _dp), _p(ld, _dp), cS, cE, int(centro_known), W, error, max_gap,
                            mu, M, threads, _p(win, _dp))
    assert rc == 0
    return win


def ref_hr2_ld(geno, W, idx=None, threads=1):
    geno = np.ascontiguousarray(geno, dtype=np.int16)
    nloci, nind = geno.shape
    if idx is None:
        idx = np.arange(nind, dtype=np.int32)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    hom = np.empty(nloci, dtype=np.float64)
    ld = np.empty((nloci, W), dtype=np.float64)
    rc = ref().ref_calcHR2LD(nloci, nind, _p(geno, _sp), W, threads, _p(idx, _ip), idx.shape[0],
                             _p(hom, _dp), _p(ld, _dp))
    assert rc == 0
    return hom, ld


def ref_r2_ld(geno, first_copy, freq, W, idx=None, threads=1):
    geno = np.ascontiguousarray(geno, dtype=np.int16)
    fc = np.ascontiguousarray(first_copy, dtype=np.uint8)
    freq = np.ascontiguousarray(freq, dtype=np.float64)
    nloci, nind = geno.shape
    if idx is None:
        idx = np.arange(nind, dtype=np.int32)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    ld = np.empty((nloci, W), dtype=np.float64)
    rc = ref().ref_calcR2LD(nloci, nind, _p(geno, _sp), _p(fc, _bp), _p(freq, _dp), W, threads,
                            _p(idx, _ip), idx.shape[0], _p(ld, _dp))
    assert rc == 0
    return ld


def ref_assemble_roh(win, pos, cS, cE, cutoff, W, max_gap, overlap_frac, centro_known=True):
    """the reference's assembleROHWindows on one chromosome -> list of (individual, start, stop)"""
    win = np.ascontiguousarray(win, dtype=np.float64)
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    nind, nloci = win.shape
    cap = nind * nloci
    ind = np.empty(cap, dtype=np.int32)
    a = np.empty(cap, dtype=np.float64)
    b = np.empty(cap, dtype=np.float64)
    n = ref().ref_assembleROH(nloci, nind, _p(win, _dp), _p(pos, _ip), cS, cE, int(centro_known), cutoff, W, max_gap,
                              overlap_frac, cap, _p(ind, _ip), _p(a, _dp), _p(b, _dp))
    assert 0 <= n <= cap
    return [(int(ind[k]), float(a[k]), float(b[k])) for k in range(n)]


def ref_flatten(win, step):
    win = np.ascontiguousarray(win, dtype=np.float64)
    nind, nloci = win.shape
    out = np.empty(win.size, dtype=np.float64)
    n = ref().ref_flatten(nloci, nind, _p(win, _dp), step, _p(out, _dp))
    assert n >= 0
    return out[:n].copy()


def bits_equal(a, b):
    """Bitwise comparison of two float64 arrays (NaN payloads and signed zeros included)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def count_mismatch(a, b):
    return int(np.count_nonzero(a.view(np.uint64) != b.view(np.uint64)))


# ---------------------------------------------------------------- small random panels

def random_panel(rng, nloci, nind, *, miss=0.03, gaps=2, max_gap=200000, centro=True,
                 mono=0.01, spacing=2000):
    """A chromosome-sized test panel with >max_gap holes, a centromere that contains SNPs,
    missing genotypes (-9) and a few freq in {0,1} survivors (reachable via --freq-file)."""
    steps = rng.integers(1, 2 * spacing, size=nloci).astype(np.int64)
    for _ in range(gaps):
        if nloci > 2:
            steps[rng.integers(1, nloci)] += max_gap + rng.integers(1, 1000)
    pos = np.cumsum(steps)
    pos = pos.astype(np.int32)
    if centro and nloci > 8:
        a = int(rng.integers(nloci // 4, nloci // 2))
        b = min(nloci - 1, a + int(rng.integers(0, 4)))
        cS, cE = int(pos[a]) - int(rng.integers(0, 3)), int(pos[b]) + int(rng.integers(0, 3))
    else:
        cS, cE = 0, 0
    freq = rng.uniform(0.02, 0.98, size=nloci)
    k = rng.random(nloci)
    freq[k < mono / 2] = 0.0
    freq[(k >= mono / 2) & (k < mono)] = 1.0
    p = freq[:, None]
    u = rng.random((nloci, nind))
    geno = np.where(u < (1 - p) ** 2, 0, np.where(u < (1 - p) ** 2 + 2 * p * (1 - p), 1, 2)).astype(np.int16)
    geno[rng.random((nloci, nind)) < miss] = -9
    return geno, freq, pos, cS, cE
