"""ctypes binding of libgarlic_hip.so (C ABI declared in include/garlic_hip.h).

This is plumbing for the tests and bench.py; the product is the shared library.  There is no
CPU fallback anywhere in this package: if the library is missing or no gfx950 device is
present, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgarlic_hip.so")

OK, ERR_INVALID, ERR_HIP, ERR_STATE, ERR_NOMEM = 0, 1, 2, 3, 4
HOST, DEVICE = 0, 1
MISSING = -9999.0

# every symbol include/garlic_hip.h declares (tests check the library exports them all)
SYMBOLS = [
    "garlic_hip_abi_version", "garlic_hip_last_error", "garlic_hip_device_count",
    "garlic_ctx_create", "garlic_ctx_destroy", "garlic_ctx_synchronize",
    "garlic_panel_create", "garlic_panel_destroy", "garlic_panel_set_map",
    "garlic_panel_set_freq", "garlic_panel_set_genotypes", "garlic_panel_set_genotypes_2bit", "garlic_panel_set_gl", "garlic_panel_set_gl_codes",
    "garlic_panel_set_phase",
    "garlic_panel_set_ld", "garlic_lod_out_layout", "garlic_lod_windows", "garlic_lod_windows_multi",
    "garlic_wlod_windows", "garlic_lod_flatten", "garlic_last_call_stats",
    "garlic_panel_compute_ld", "garlic_ld_counts", "garlic_ld_finish", "garlic_roh_coverage",
    "garlic_panel_release_scratch",
    "garlic_lod_feed", "garlic_ctx_set_async",
    "garlic_recent_kernel_ms", "garlic_panel_tgls_mode", "garlic_lod_feed_subset", "garlic_lod_feed_multi",
    "garlic_device_alloc", "garlic_device_free", "garlic_panel_chain_kind", "garlic_device_alloc_stats",
    "garlic_panel_alloc_scores", "garlic_device_trim", "garlic_roh_coverage_fused", "garlic_roh_segments",
    "garlic_panel_alloc_scores_info",
]


class CallStats(C.Structure):
    _fields_ = [("n_segments", C.c_int64), ("n_runs", C.c_int64), ("n_chain_items", C.c_int64),
                ("n_valid_windows", C.c_int64), ("n_missing", C.c_int64),
                ("chain_kernel_ms", C.c_float), ("total_ms", C.c_float),
                ("n_stall_reruns", C.c_int64), ("n_count_timeouts", C.c_int64)]


class GarlicError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libgarlic_hip error {code}: {msg}")
        self.code = code


_lib = None
_vp = C.c_void_p
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)


ABI_VERSION = 8   # GARLIC_HIP_ABI_VERSION of include/garlic_hip.h these bindings were written against


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C garlic_amd/csrc` (there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    L.garlic_hip_abi_version.restype = C.c_int
    if L.garlic_hip_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {L.garlic_hip_abi_version()}, these bindings need "
                          f"{ABI_VERSION}: rebuild it (make -C garlic_amd/csrc)")
    L.garlic_hip_last_error.restype = C.c_char_p
    L.garlic_hip_device_count.argtypes = [_i32p]
    L.garlic_ctx_create.argtypes = [C.c_int32, _vp, C.POINTER(_vp)]
    L.garlic_ctx_destroy.argtypes = [_vp]
    L.garlic_ctx_synchronize.argtypes = [_vp]
    L.garlic_ctx_set_async.argtypes = [_vp, C.c_int32]
    L.garlic_device_alloc.argtypes = [_vp, C.c_int64, C.POINTER(_vp)]
    L.garlic_device_alloc_stats.argtypes = [_vp, _i64p, _i64p, _i64p]
    L.garlic_panel_alloc_scores.argtypes = [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.POINTER(_vp),
                                            C.POINTER(C.c_float)]
    L.garlic_device_free.argtypes = [_vp, _vp]
    L.garlic_recent_kernel_ms.argtypes = [_vp, C.POINTER(C.c_float), C.c_int32, _i32p]
    L.garlic_panel_create.argtypes = [_vp, C.c_int32, _i32p, C.c_int32, C.POINTER(_vp)]
    L.garlic_panel_destroy.argtypes = [_vp]
    L.garlic_panel_set_map.argtypes = [_vp, _i32p, _f64p, _i32p, _i32p]
    L.garlic_panel_set_freq.argtypes = [_vp, _f64p]
    L.garlic_panel_set_genotypes.argtypes = [_vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32]
    L.garlic_panel_set_gl.argtypes = [_vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32]
    L.garlic_panel_release_scratch.argtypes = [_vp]
    L.garlic_lod_windows_multi.argtypes = [_vp, _i32p, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_int32, _vp, C.c_int64, C.c_int32]
    L.garlic_panel_set_gl_codes.argtypes = [_vp, _vp, C.c_int64, C.c_int64, C.c_int64, _vp, C.c_int32, C.c_int32]
    L.garlic_panel_set_genotypes_2bit.argtypes = [_vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32]
    L.garlic_panel_set_phase.argtypes = [_vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32]
    L.garlic_panel_set_ld.argtypes = [_vp, C.c_int32, _vp, C.c_int32]
    L.garlic_lod_out_layout.argtypes = [_vp, C.c_int32, C.c_int32, _i64p, _i64p, _i64p]
    L.garlic_lod_windows.argtypes = [_vp, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, _vp, C.c_int32]
    L.garlic_wlod_windows.argtypes = [_vp, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_double, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_int32]
    L.garlic_lod_flatten.argtypes = [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_int64, _i64p]
    L.garlic_last_call_stats.argtypes = [_vp, C.POINTER(CallStats)]
    L.garlic_panel_compute_ld.argtypes = [_vp, C.c_int32, C.c_int32, _i32p, C.c_int32, _vp, C.c_int32]
    L.garlic_ld_counts.argtypes = [_vp, C.c_int32, C.c_int32, _i32p, C.c_int32, _vp, _vp, C.c_int32]
    L.garlic_ld_finish.argtypes = [_vp, C.c_int32, C.c_int32, _vp, _vp, _vp, C.c_int32]
    L.garlic_lod_feed.argtypes = [_vp, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_double, C.c_int32, _vp, C.c_int64, _i64p, _i64p]
    L.garlic_roh_coverage.argtypes = [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_double, _vp, C.c_int32,
                                      C.c_int32]
    L.garlic_roh_coverage_fused.argtypes = [_vp, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                            C.c_double, _vp, C.c_int32, C.c_int32]
    L.garlic_roh_segments.argtypes = [_vp, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                      C.c_double, C.c_double, _vp, C.c_int64, _i64p]
    L.garlic_lod_feed_subset.argtypes = [_vp, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_double, C.c_int32, _i32p, C.c_int32, _vp, C.c_int64, _i64p, _i64p]
    L.garlic_lod_feed_multi.argtypes = [_vp, _i32p, _i32p, C.c_int32, C.c_double, C.c_int32, _i32p, C.c_int32,
                                        C.POINTER(C.c_void_p), _i64p, _i64p, _i64p]
    L.garlic_panel_tgls_mode.argtypes = [_vp, _i32p, _i32p]
    L.garlic_panel_chain_kind.argtypes = [_vp, _i32p]
    for name in SYMBOLS:
        f = getattr(L, name)
        if f.restype is C.c_int and name not in ("garlic_hip_abi_version",):
            f.restype = C.c_int
    _lib = L
    return L


def check(rc):
    if rc != OK:
        raise GarlicError(rc, lib().garlic_hip_last_error().decode())


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)


class Context:
    """One device + one HIP stream (garlic_ctx)."""

    def __init__(self, device=0, stream=None):
        self.handle = _vp()
        check(lib().garlic_ctx_create(device, _vp(stream) if stream else None, C.byref(self.handle)))
        self.device = device

    def synchronize(self):
        check(lib().garlic_ctx_synchronize(self.handle))

    def recent_kernel_ms(self, n=32):
        """HIP-event durations of the dominant kernel of the last <= n (<= 32) score calls, oldest first"""
        buf = (C.c_float * n)()
        got = C.c_int32()
        check(lib().garlic_recent_kernel_ms(self.handle, buf, n, C.byref(got)))
        return [float(buf[i]) for i in range(got.value)]

    def set_async(self, on=True):
        """device-output calls that repeat the previous call's arguments only enqueue (see garlic_hip.h)"""
        check(lib().garlic_ctx_set_async(self.handle, int(on)))

    def alloc_scores(self, n_doubles):
        """device memory for a score matrix through garlic_device_alloc (pooled; see garlic_hip.h)"""
        return DeviceBuffer(self, int(n_doubles) * 8)

    def trim(self):
        """garlic_device_trim: idle pooled score buffers give their memory back"""
        check(lib().garlic_device_trim(self.handle))

    def alloc_stats(self):
        """garlic_device_alloc_stats: (live, pooled, reserved) bytes of score memory on this context's device"""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib().garlic_device_alloc_stats(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def close(self):
        if self.handle:
            lib().garlic_ctx_destroy(self.handle)
            self.handle = _vp()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class DeviceBuffer:
    """garlic_device_alloc / garlic_device_free; `.ptr` for the device-output calls, `.tensor()` to look at it"""

    def __init__(self, ctx, nbytes, ptr=None):
        self.ctx, self.nbytes = ctx, nbytes
        if ptr is None:
            p = _vp()
            check(lib().garlic_device_alloc(ctx.handle, nbytes, C.byref(p)))
            ptr = p.value
        self.ptr = ptr

    def data_ptr(self):
        return self.ptr

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.nbytes // 8,), "typestr": "<f8", "data": (self.ptr, False), "version": 2, "strides": None}

    def tensor(self):
        """a float64 torch tensor over the same memory (valid while this object lives)"""
        import torch
        return torch.as_tensor(self, device=f"cuda:{self.ctx.device}")

    def free(self):
        if self.ptr:
            check(lib().garlic_device_free(self.ctx.handle, _vp(self.ptr)))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Panel:
    """Device-resident genotype panel (garlic_panel) for the individuals one context owns."""

    def __init__(self, ctx, chr_nloci, nind):
        self.ctx = ctx
        self.chr_nloci = np.ascontiguousarray(chr_nloci, dtype=np.int32)
        self.nchr = int(self.chr_nloci.shape[0])
        self.nloci = int(self.chr_nloci.sum())
        self.nind = int(nind)
        self.handle = _vp()
        check(lib().garlic_panel_create(ctx.handle, self.nchr, _ptr(self.chr_nloci, _i32p), self.nind,
                                        C.byref(self.handle)))

    def close(self):
        if self.handle:
            lib().garlic_panel_destroy(self.handle)
            self.handle = _vp()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_map(self, pos, centro_start, centro_end, gpos=None):
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        cs = np.ascontiguousarray(centro_start, dtype=np.int32)
        ce = np.ascontiguousarray(centro_end, dtype=np.int32)
        assert pos.shape[0] == self.nloci and cs.shape[0] == self.nchr and ce.shape[0] == self.nchr
        if gpos is not None:
            gpos = np.ascontiguousarray(gpos, dtype=np.float64)
            assert gpos.shape[0] == self.nloci
        check(lib().garlic_panel_set_map(self.handle, _ptr(pos, _i32p), _ptr(gpos, _f64p),
                                         _ptr(cs, _i32p), _ptr(ce, _i32p)))

    def set_freq(self, freq):
        freq = np.ascontiguousarray(freq, dtype=np.float64)
        assert freq.shape[0] == self.nloci
        check(lib().garlic_panel_set_freq(self.handle, _ptr(freq, _f64p)))

    def set_genotypes(self, geno, locus_begin=0):
        """geno: int16 numpy array [nloci_chunk][>= nind] (host)."""
        geno = np.asarray(geno)
        assert geno.dtype == np.int16 and geno.ndim == 2 and geno.strides[1] == 2
        ld = geno.strides[0] // 2
        check(lib().garlic_panel_set_genotypes(self.handle, _vp(geno.ctypes.data), ld, locus_begin,
                                               geno.shape[0], HOST))

    def set_genotypes_device(self, ptr, ld, locus_begin, locus_count):
        """ptr: device address of int16 [locus_count][ld] (e.g. torch tensor .data_ptr())."""
        check(lib().garlic_panel_set_genotypes(self.handle, _vp(ptr), ld, locus_begin, locus_count,
                                               DEVICE))

    def set_gl(self, gl, locus_begin=0):
        """gl: float64 numpy array [nloci_chunk][>= nind]: per-genotype error probabilities (TGLS)."""
        gl = np.asarray(gl)
        assert gl.dtype == np.float64 and gl.ndim == 2 and gl.strides[1] == 8
        check(lib().garlic_panel_set_gl(self.handle, _vp(gl.ctypes.data), gl.strides[0] // 8, locus_begin,
                                        gl.shape[0], HOST))

    def set_genotypes_2bit(self, rows, ind_offset=0, locus_begin=0):
        """rows: uint8 [nloci_chunk][row_bytes], 4 genotypes per byte (3 = missing), individuals of the
        whole data set; this panel's individuals start at ind_offset."""
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        assert rows.ndim == 2
        check(lib().garlic_panel_set_genotypes_2bit(self.handle, _vp(rows.ctypes.data), rows.shape[1], ind_offset,
                                                    locus_begin, rows.shape[0], HOST))

    def set_gl_codes(self, codes, values, locus_begin=0):
        """codes: uint8 [nloci_chunk][nind] indexing values (float64, <= 256 error probabilities)"""
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        values = np.ascontiguousarray(values, dtype=np.float64)
        check(lib().garlic_panel_set_gl_codes(self.handle, _vp(codes.ctypes.data), codes.shape[1], locus_begin,
                                              codes.shape[0], _vp(values.ctypes.data), values.shape[0], HOST))

    def release_scratch(self):
        """free the device scratch the panel keeps between calls (LD buffers, score / feed scratch)"""
        check(lib().garlic_panel_release_scratch(self.handle))

    def set_phase(self, first_copy, locus_begin=0):
        """first_copy: uint8/bool [nloci_chunk][nind], HapData::firstCopy (--phased)."""
        fc = np.ascontiguousarray(first_copy).view(np.uint8) if np.asarray(first_copy).dtype == np.bool_ \
            else np.ascontiguousarray(first_copy, dtype=np.uint8)
        assert fc.ndim == 2 and fc.shape[1] >= self.nind
        check(lib().garlic_panel_set_phase(self.handle, _vp(fc.ctypes.data), fc.shape[1], locus_begin,
                                           fc.shape[0], HOST))

    def set_gl_device(self, ptr, ld, locus_begin, locus_count):
        """ptr: device address of float64 [locus_count][ld] per-genotype error probabilities."""
        check(lib().garlic_panel_set_gl(self.handle, _vp(ptr), ld, locus_begin, locus_count, DEVICE))

    def set_ld_device(self, winsize, ptr):
        """ptr: device address of float64 [nloci][winsize] LD weights."""
        check(lib().garlic_panel_set_ld(self.handle, winsize, _vp(ptr), DEVICE))

    def wlod_windows_device(self, out_ptr, winsize, error, max_gap, M, mu, ind_begin=0, ind_count=None,
                            pitch_align=32, use_gl=False):
        ind_count = self.nind - ind_begin if ind_count is None else ind_count
        check(lib().garlic_wlod_windows(self.handle, winsize, error, max_gap, int(use_gl), M, mu, ind_begin,
                                        ind_count, pitch_align, _vp(out_ptr), DEVICE))

    def set_ld(self, winsize, ld):
        """ld: float64 [nloci][winsize] LD weights of wLOD (all chromosomes concatenated)."""
        ld = np.ascontiguousarray(ld, dtype=np.float64)
        assert ld.shape == (self.nloci, winsize)
        check(lib().garlic_panel_set_ld(self.handle, winsize, _vp(ld.ctypes.data), HOST))

    @staticmethod
    def _sub(sub_idx):
        """None = every individual; an array = exactly those, an EMPTY array = none of them (a shard
        that holds no member of a panel-wide subsample): the pointer must then still be non-NULL"""
        if sub_idx is None:
            return None, 0
        sub = np.ascontiguousarray(sub_idx, dtype=np.int32)
        n = int(sub.shape[0])
        if n == 0:
            sub = np.zeros(1, dtype=np.int32)
        return sub, n

    def compute_ld(self, winsize, sub_idx=None, want_output=True, phased=False):
        """calcHR2LD (phased: calcR2LD) on the device (sub_idx: the --ld-subsample individuals,
        None = all); installs the weights for wlod_windows and returns them as float64
        [nloci][winsize]."""
        sub, n = self._sub(sub_idx)
        out = np.empty((self.nloci, winsize), dtype=np.float64) if want_output else None
        check(lib().garlic_panel_compute_ld(self.handle, winsize, int(phased), _ptr(sub, _i32p), n,
                                            _vp(out.ctypes.data) if want_output else None, HOST))
        return out

    def ld_counts(self, winsize, sub_idx=None, phased=False):
        """Integer part of the LD weights for this shard's individuals: (locus_counts [nloci][2],
        pair_counts [nloci][winsize][2]) int32; sum over shards, then ld_finish."""
        sub, n = self._sub(sub_idx)
        loc = np.empty((self.nloci, 2), dtype=np.int32)
        pair = np.empty((self.nloci, winsize, 2), dtype=np.int32)
        check(lib().garlic_ld_counts(self.handle, winsize, int(phased), _ptr(sub, _i32p), n, _vp(loc.ctypes.data),
                                     _vp(pair.ctypes.data), HOST))
        return loc, pair

    def ld_finish(self, winsize, locus_counts, pair_counts, want_output=True, phased=False):
        loc = np.ascontiguousarray(locus_counts, dtype=np.int32)
        pair = np.ascontiguousarray(pair_counts, dtype=np.int32)
        assert loc.shape == (self.nloci, 2) and pair.shape == (self.nloci, winsize, 2)
        out = np.empty((self.nloci, winsize), dtype=np.float64) if want_output else None
        check(lib().garlic_ld_finish(self.handle, winsize, int(phased), _vp(loc.ctypes.data), _vp(pair.ctypes.data),
                                     _vp(out.ctypes.data) if want_output else None, HOST))
        return out

    def ld_counts_device(self, winsize, locus_ptr, pair_ptr, sub_idx=None, phased=False):
        sub, n = self._sub(sub_idx)
        check(lib().garlic_ld_counts(self.handle, winsize, int(phased), _ptr(sub, _i32p), n, _vp(locus_ptr),
                                     _vp(pair_ptr), DEVICE))

    def ld_finish_device(self, winsize, locus_ptr, pair_ptr, ld_ptr=None, phased=False):
        check(lib().garlic_ld_finish(self.handle, winsize, int(phased), _vp(locus_ptr), _vp(pair_ptr),
                                     _vp(ld_ptr) if ld_ptr else None, DEVICE))

    def out_layout(self, pitch_align=1, nind_out=None):
        nind_out = self.nind if nind_out is None else nind_out
        base = np.empty(self.nchr, dtype=np.int64)
        pitch = np.empty(self.nchr, dtype=np.int64)
        total = C.c_int64()
        check(lib().garlic_lod_out_layout(self.handle, pitch_align, nind_out, _ptr(base, _i64p),
                                          _ptr(pitch, _i64p), C.byref(total)))
        return base, pitch, total.value

    def lod_windows(self, winsize, error, max_gap, ind_begin=0, ind_count=None, pitch_align=1,
                    use_gl=False):
        """Host-output convenience: returns a list of per-chromosome [ind_count][nloci_c] arrays."""
        ind_count = self.nind - ind_begin if ind_count is None else ind_count
        base, pitch, total = self.out_layout(pitch_align, ind_count)
        out = np.empty(total, dtype=np.float64)
        check(lib().garlic_lod_windows(self.handle, winsize, error, max_gap, int(use_gl), ind_begin,
                                       ind_count, pitch_align, _vp(out.ctypes.data), HOST))
        res = []
        for c in range(self.nchr):
            n = int(self.chr_nloci[c])
            blk = out[base[c]: base[c] + ind_count * pitch[c]].reshape(ind_count, pitch[c])
            res.append(blk[:, :n])
        return res

    def wlod_windows(self, winsize, error, max_gap, M, mu, ind_begin=0, ind_count=None, pitch_align=1,
                     use_gl=False):
        ind_count = self.nind - ind_begin if ind_count is None else ind_count
        base, pitch, total = self.out_layout(pitch_align, ind_count)
        out = np.empty(total, dtype=np.float64)
        check(lib().garlic_wlod_windows(self.handle, winsize, error, max_gap, int(use_gl), M, mu, ind_begin,
                                        ind_count, pitch_align, _vp(out.ctypes.data), HOST))
        res = []
        for c in range(self.nchr):
            n = int(self.chr_nloci[c])
            blk = out[base[c]: base[c] + ind_count * pitch[c]].reshape(ind_count, pitch[c])
            res.append(blk[:, :n])
        return res

    def lod_windows_multi(self, winsizes, error, max_gap, pitch_align=1, use_gl=False):
        """Host-output convenience for several window sizes: {W: [per-chromosome [nind][nloci_c] arrays]}."""
        ws = np.ascontiguousarray(winsizes, dtype=np.int32)
        base, pitch, total = self.out_layout(pitch_align, self.nind)
        out = np.empty((len(ws), total), dtype=np.float64)
        check(lib().garlic_lod_windows_multi(self.handle, _ptr(ws, _i32p), len(ws), error, max_gap, int(use_gl), 0,
                                             self.nind, pitch_align, _vp(out.ctypes.data), total, HOST))
        res = {}
        for k, W in enumerate(ws):
            res[int(W)] = [out[k, base[c]: base[c] + self.nind * pitch[c]].reshape(self.nind, pitch[c])[:, :int(self.chr_nloci[c])]
                           for c in range(self.nchr)]
        return res

    def lod_windows_device(self, out_ptr, winsize, error, max_gap, ind_begin=0, ind_count=None,
                           pitch_align=32, use_gl=False):
        ind_count = self.nind - ind_begin if ind_count is None else ind_count
        check(lib().garlic_lod_windows(self.handle, winsize, error, max_gap, int(use_gl), ind_begin,
                                       ind_count, pitch_align, _vp(out_ptr), DEVICE))

    def flatten_device(self, scores_ptr, step, feed_ptr, feed_capacity, pitch_align=32, nind_out=None):
        """KDE feed (convertWinData2DoubleData) of device-resident scores; returns the count."""
        nind_out = self.nind if nind_out is None else nind_out
        n = C.c_int64()
        check(lib().garlic_lod_flatten(self.handle, _vp(scores_ptr), pitch_align, nind_out, step,
                                       _vp(feed_ptr) if feed_ptr else None, feed_capacity, C.byref(n)))
        return n.value

    def lod_feed(self, winsize, error, max_gap, step, use_gl=False, weighted=False, M=7, mu=1e-9, copy=True,
                 ind_idx=None):
        """scores + thinning on the device: returns (feed float64 [count], per-chromosome counts).
        copy=False: the feed is a view of a buffer the panel object reuses for the next call.
        ind_idx: only these individuals, in this order (convertSubsetWinData2DoubleData, --kde-subsample)"""
        idx = None if ind_idx is None else np.ascontiguousarray(ind_idx, dtype=np.int32)
        n_rows = self.nind if idx is None else int(idx.shape[0])
        cap = int(sum((int(n) + step - 1) // step for n in self.chr_nloci)) * n_rows
        if copy:
            feed = np.empty(max(cap, 1), dtype=np.float64)
        else:
            if getattr(self, "_feed_buf", None) is None or self._feed_buf.shape[0] < max(cap, 1):
                self._feed_buf = np.empty(max(cap, 1), dtype=np.float64)
            feed = self._feed_buf
        n = C.c_int64()
        per_chr = np.zeros(self.nchr, dtype=np.int64)
        check(lib().garlic_lod_feed_subset(self.handle, winsize, error, max_gap, int(use_gl), int(weighted), M, mu,
                                           step, _ptr(idx, _i32p), 0 if idx is None else n_rows,
                                           _vp(feed.ctypes.data), cap, C.byref(n), _ptr(per_chr, _i64p)))
        return (feed[: n.value].copy() if copy else feed[: n.value]), per_chr

    def lod_feed_multi(self, winsizes, error, max_gap, steps=None, ind_idx=None, copy=True):
        """garlic_lod_feed_multi: the feeds of several window sizes in one call (unweighted --error scores; steps
        default to the window sizes).  Returns ([feed per size], per-chromosome counts [n_sizes, nchr]).
        copy=False: the feeds are views of buffers the panel object reuses for the next call."""
        sizes = np.ascontiguousarray(winsizes, dtype=np.int32)
        st = sizes.copy() if steps is None else np.ascontiguousarray(steps, dtype=np.int32)
        idx = None if ind_idx is None else np.ascontiguousarray(ind_idx, dtype=np.int32)
        n_rows = self.nind if idx is None else int(idx.shape[0])
        caps = np.array([max(1, int(sum((int(n) + int(s) - 1) // int(s) for n in self.chr_nloci)) * n_rows) for s in st],
                        dtype=np.int64)
        if copy:
            bufs = [np.empty(int(c), dtype=np.float64) for c in caps]
        else:
            old = getattr(self, "_feed_multi", [])
            bufs = [old[i] if i < len(old) and old[i].shape[0] >= int(c) else np.empty(int(c), dtype=np.float64)
                    for i, c in enumerate(caps)]
            self._feed_multi = bufs
        ptrs = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
        counts = np.zeros(len(bufs), dtype=np.int64)
        per_chr = np.zeros((len(bufs), self.nchr), dtype=np.int64)
        check(lib().garlic_lod_feed_multi(self.handle, _ptr(sizes, _i32p), _ptr(st, _i32p), len(bufs), error, max_gap,
                                          _ptr(idx, _i32p), 0 if idx is None else n_rows, ptrs, _ptr(caps, _i64p),
                                          _ptr(counts, _i64p), _ptr(per_chr, _i64p)))
        return [b[: int(n)] for b, n in zip(bufs, counts)], per_chr

    def alloc_scores(self, winsize, error, max_gap, pitch_align=32, nind_out=None, candidates=0):
        """garlic_panel_alloc_scores: score memory in the chain kernel's fast placement (the real kernel timed into
        `candidates` buffers, the fastest kept).  Returns (DeviceBuffer, [kernel ms per candidate])."""
        nind_out = self.nind if nind_out is None else nind_out
        n = candidates if candidates > 0 else 4
        ms = (C.c_float * n)()
        ptr = _vp()
        check(lib().garlic_panel_alloc_scores(self.handle, pitch_align, nind_out, winsize, error, max_gap, candidates,
                                              C.byref(ptr), ms))
        _, _, total = self.out_layout(pitch_align, nind_out)
        return DeviceBuffer(self.ctx, int(total) * 8, ptr=ptr.value), [float(x) for x in ms]

    def alloc_scores_info(self):
        """garlic_panel_alloc_scores_info: what the last alloc_scores on this panel drew"""
        drawn, rounds, reached = C.c_int32(), C.c_int32(), C.c_int32()
        best, med, worst, target = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        check(lib().garlic_panel_alloc_scores_info(self.handle, C.byref(drawn), C.byref(rounds), C.byref(best), C.byref(med),
                                                   C.byref(worst), C.byref(target), C.byref(reached)))
        return {"candidates_drawn": drawn.value, "rounds": rounds.value, "best_ms": best.value, "median_ms": med.value,
                "worst_ms": worst.value, "target_ms_at_0.74_of_hbm": target.value, "reached_target": bool(reached.value)}

    def chain_kind(self):
        """0 tuned chain, 1 tuned chain + scan for the value -9999.0 (none found), 2 by-value chain (garlic_hip.h)"""
        k = C.c_int32()
        check(lib().garlic_panel_chain_kind(self.handle, C.byref(k)))
        return k.value

    def tgls_mode(self):
        """(mode, terms_by): mode 0 none / 1 dictionary codes / 2 continuous values; terms_by 0 tabulated or
        nothing yet / 1 device log10 / 2 host libm"""
        mode, by = C.c_int32(), C.c_int32()
        check(lib().garlic_panel_tgls_mode(self.handle, C.byref(mode), C.byref(by)))
        return mode.value, by.value

    def roh_coverage(self, scores_ptr, winsize, cutoff, pitch_align=32, nind_out=None, inwin_pitch_align=1):
        """assembleROHWindows' coverage counts of device-resident scores: list of per-chromosome int16
        [nind_out][pitch_c] host arrays (pitch_c = nloci_c rounded up to inwin_pitch_align; a multiple of 8 lets the kernel
        store 16 bytes at a time)."""
        nind_out = self.nind if nind_out is None else nind_out
        base, pitch, total = self.out_layout(inwin_pitch_align, nind_out)
        out = np.empty(total, dtype=np.int16)
        check(lib().garlic_roh_coverage(self.handle, _vp(scores_ptr), pitch_align, nind_out, winsize, cutoff,
                                        _vp(out.ctypes.data), inwin_pitch_align, HOST))
        return [out[base[c]: base[c] + nind_out * pitch[c]].reshape(nind_out, pitch[c])
                for c in range(self.nchr)]

    def roh_coverage_fused(self, winsize, error, max_gap, cutoff, pitch_align=1, use_gl=False, weighted=False, M=7, mu=1e-9):
        """coverage counts without computing the score matrix (unweighted --error scores, or --weighted with or without
        likelihoods): list of per-chromosome int16 [nind][pitch_c] host arrays (pitch_c = nloci_c rounded up to pitch_align)"""
        base, pitch, total = self.out_layout(pitch_align, self.nind)
        out = np.empty(total, dtype=np.int16)
        check(lib().garlic_roh_coverage_fused(self.handle, winsize, error, max_gap, int(use_gl), int(weighted), M, mu, cutoff,
                                              _vp(out.ctypes.data), pitch_align, HOST))
        return [out[base[c]: base[c] + self.nind * pitch[c]].reshape(self.nind, pitch[c]) for c in range(self.nchr)]

    def roh_segments(self, winsize, error, max_gap, cutoff, overlap_frac, use_gl=False, weighted=False, M=7, mu=1e-9, capacity=None):
        """garlic_roh_segments: the ROH segments of assembleROHWindows without scores or counts in memory -> int32 array
        [n][4] of (individual, chromosome, first SNP, last SNP), in the reference's order.  capacity None: room for 64 per
        individual, once more with the number found if that was too little."""
        n = C.c_int64()
        args = (self.handle, winsize, error, max_gap, int(use_gl), int(weighted), M, mu, cutoff, overlap_frac)
        cap = max(1024, 64 * self.nind) if capacity is None else int(capacity)
        for _ in range(2):
            out = np.empty((max(cap, 1), 4), dtype=np.int32)
            check(lib().garlic_roh_segments(*args, _vp(out.ctypes.data), cap, C.byref(n)))
            if n.value <= cap or capacity is not None:
                break
            cap = n.value           # (nothing usable was written: once more, with room)
        if n.value > cap:
            raise GarlicError(1, f"garlic_roh_segments: {n.value} segments, room for {cap}")
        return out[:n.value].copy()

    def roh_coverage_fused_device(self, winsize, error, max_gap, cutoff, out_ptr, pitch_align=8, use_gl=False, weighted=False,
                                  M=7, mu=1e-9):
        """the same counts into device memory (int16 rows: out_layout(pitch_align, nind))"""
        check(lib().garlic_roh_coverage_fused(self.handle, winsize, error, max_gap, int(use_gl), int(weighted), M, mu, cutoff,
                                              _vp(out_ptr), pitch_align, DEVICE))

    def roh_coverage_device(self, scores_ptr, winsize, cutoff, out_ptr, pitch_align=32, nind_out=None, inwin_pitch_align=1):
        """the same counts into device memory (int16 rows: out_layout(inwin_pitch_align, nind_out))"""
        nind_out = self.nind if nind_out is None else nind_out
        check(lib().garlic_roh_coverage(self.handle, _vp(scores_ptr), pitch_align, nind_out, winsize, cutoff,
                                        _vp(out_ptr), inwin_pitch_align, DEVICE))

    def stats(self):
        st = CallStats()
        check(lib().garlic_last_call_stats(self.handle, C.byref(st)))
        return {k: getattr(st, k) for k, _ in CallStats._fields_}
