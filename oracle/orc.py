"""ORACLE — TEST INFRASTRUCTURE ONLY (ctypes binding of oracle/liboracle.so and oracle/_ref/libref.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package kmer_index_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MODE_FAITHFUL, MODE_INTENDED = 0, 1
ST_OK, ST_TOO_LONG, ST_FANOUT, ST_EMPTY_QUERY = 0, 1, 2, 3

_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build(ref: bool = True) -> None:
    """Compile the oracle (and, when /root/reference exists, oracle/_ref)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    if ref and os.path.isdir(os.environ.get("KMX_REFERENCE", "/root/reference")):
        subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.orc_fast_pow.restype = C.c_uint64
        L.orc_fast_pow.argtypes = [C.c_uint64, C.c_uint8]
        L.orc_choose_best_k.argtypes = [_u64p, C.c_uint64, C.c_uint32, _u32p]
        L.orc_bitset_words.restype = C.c_int64
        L.orc_bitset_words.argtypes = [C.c_uint64, C.c_int, _u64p, C.c_uint64, _u64p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.orc_plan.restype = C.c_uint64
        L.orc_plan.argtypes = [_u32p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.orc_build.restype = C.c_void_p
        L.orc_build.argtypes = [_u8p, C.c_uint64, C.c_uint32, _u32p, C.c_uint32, C.c_uint32]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_free_buf.argtypes = [C.c_void_p]
        L.orc_search.restype = C.c_int
        L.orc_search.argtypes = [C.c_void_p, _u8p, C.c_uint64, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
        L.orc_search_batch.restype = C.c_int
        L.orc_search_batch.argtypes = [C.c_void_p, _u8p, _u64p, C.c_uint64, C.c_int, C.c_uint32, C.c_int,
                                       C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_uint64)]
        L.orc_search_batch_on.restype = C.c_int
        L.orc_search_batch_on.argtypes = L.orc_search_batch.argtypes + [C.c_void_p]
        L.orc_naive_scan.restype = C.c_uint64
        L.orc_naive_scan.argtypes = [_u8p, C.c_uint64, _u8p, C.c_uint64, _u32p, C.c_uint64]
        L.orc_naive_batch.restype = C.c_uint64
        L.orc_naive_batch.argtypes = [_u8p, C.c_uint64, _u8p, _u64p, C.c_uint64, _u64p, C.POINTER(C.c_void_p)]
        _lib = L
    return _lib


def ref_lib():
    """The real reference's fast_pow / compressed_bitset / thread_pool, or None when not built."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libref.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_fast_pow.restype = C.c_uint64
        R.ref_fast_pow.argtypes = [C.c_uint64, C.c_uint8]
        R.ref_bitset_words.restype = C.c_int64
        R.ref_bitset_words.argtypes = [C.c_uint64, C.c_int, _u64p, C.c_uint64, _u64p, C.c_uint64, C.POINTER(C.c_uint64)]
        R.ref_pool_sum.restype = C.c_uint64
        R.ref_pool_sum.argtypes = [C.c_uint32, C.c_uint32]
        _ref = R
    return _ref


def _take(ptr, n, dtype):
    """Copy n items out of a malloc'd buffer and free it."""
    if not ptr:
        return np.zeros(0, dtype)
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(max(int(n), 1),))[: int(n)].copy()
    lib().orc_free_buf(ptr)
    return arr


def fast_pow(base: int, exp: int) -> int:
    return int(lib().orc_fast_pow(base, exp))


def choose_best_k(lengths, n_k=4):
    lengths = np.ascontiguousarray(lengths, np.uint64)
    out = np.zeros(n_k, np.uint32)
    lib().orc_choose_best_k(lengths if lengths.size else np.zeros(1, np.uint64), lengths.size, n_k, out)
    return out.tolist()


def bitset_words(n_bits, fill, ops, which="orc"):
    """ops: iterable of (index, value).  Returns (words, count_ones) or None on out_of_range."""
    enc = np.array([(int(i) << 1) | int(bool(v)) for i, v in ops], dtype=np.uint64)
    if enc.size == 0:
        enc = np.zeros(0, np.uint64)
    cap = n_bits // 64 + 2
    words = np.zeros(cap, np.uint64)
    ones = C.c_uint64(0)
    fn = lib().orc_bitset_words if which == "orc" else ref_lib().ref_bitset_words
    n = fn(n_bits, int(bool(fill)), enc, enc.size, words, cap, C.byref(ones))
    if n < 0:
        return None
    return words[:n].copy(), int(ones.value)


def plan(ks, rng=10000):
    """Planner tables: (multi[rng] bool, list-of-lists nk_sum)."""
    ks = np.ascontiguousarray(ks, np.uint32)
    multi = np.zeros(rng, np.uint8)
    off = np.zeros(rng + 1, np.uint64)
    total = lib().orc_plan(ks, ks.size, rng, multi.ctypes.data, off.ctypes.data, None, 0)
    flat = np.zeros(max(int(total), 1), np.uint32)
    lib().orc_plan(ks, ks.size, rng, multi.ctypes.data, off.ctypes.data, flat.ctypes.data, int(total))
    return multi.astype(bool), [flat[int(off[q]):int(off[q + 1])].tolist() for q in range(rng)]


class Index:
    """The restated kmer_index (one kmer_index_element per k + planner)."""

    def __init__(self, ranks, sigma, ks, n_threads=1):
        self.text = np.ascontiguousarray(ranks, np.uint8)
        self.ks = np.ascontiguousarray(ks, np.uint32)
        if self.text.size < int(self.ks.max()):
            raise ValueError("text shorter than k")
        self.sigma = int(sigma)
        self._h = lib().orc_build(self.text, self.text.size, self.sigma, self.ks, self.ks.size, n_threads)

    def close(self):
        if self._h:
            lib().orc_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def search(self, q, mode=MODE_INTENDED, want_mask=False):
        """search(q).to_vector() -> (status, positions[, mask dict])."""
        q = np.ascontiguousarray(q, np.uint8)
        pos = C.c_void_p()
        n = C.c_uint64()
        mw = C.c_void_p()
        mb = C.c_uint64()
        byp = C.c_int()
        nc = C.c_uint64()
        qq = q if q.size else np.zeros(1, np.uint8)
        st = lib().orc_search(self._h, qq, q.size, mode, C.byref(pos), C.byref(n), C.byref(mw), C.byref(mb), C.byref(byp), C.byref(nc))
        out = _take(pos.value, n.value, np.uint32)
        if not want_mask:
            return st, out
        words = _take(mw.value, mb.value // 64 + 1, np.uint64) if mw.value else np.zeros(0, np.uint64)
        return st, out, {"words": words, "bits": int(mb.value), "bypass": bool(byp.value), "candidates": int(nc.value)}

    def search_batch(self, qranks, qoff, mode=MODE_INTENDED, n_threads=1, keep_hits=True, reference_pool=False):
        """Thread-pool batch.  Returns (hit_off, positions, status, checksum).
        reference_pool: carry the chunk tasks on the REFERENCE's own thread_pool (oracle/_ref, built from its
        thread_pool.{hpp,cpp}) instead of the restated pool; raises when oracle/_ref is not built."""
        qranks = np.ascontiguousarray(qranks, np.uint8)
        qoff = np.ascontiguousarray(qoff, np.uint64)
        nq = qoff.size - 1
        hit_off = np.zeros(nq + 1, np.uint64)
        status = np.zeros(max(nq, 1), np.int32)
        pos = C.c_void_p()
        cks = C.c_uint64()
        qq = qranks if qranks.size else np.zeros(1, np.uint8)
        if reference_pool:
            R = ref_lib()
            if R is None or not hasattr(R, "ref_pool_run"):
                raise RuntimeError("oracle/_ref/libref.so (the reference's thread_pool) is not built")
            lib().orc_search_batch_on(self._h, qq, qoff, nq, mode, n_threads, int(keep_hits), hit_off.ctypes.data,
                                      C.byref(pos), status.ctypes.data, C.byref(cks), C.cast(R.ref_pool_run, C.c_void_p))
        else:
            lib().orc_search_batch(self._h, qq, qoff, nq, mode, n_threads, int(keep_hits), hit_off.ctypes.data,
                                   C.byref(pos), status.ctypes.data, C.byref(cks))
        positions = _take(pos.value, hit_off[nq], np.uint32) if keep_hits else np.zeros(0, np.uint32)
        return hit_off, positions, status[:nq], int(cks.value)


def naive_scan(text, q):
    text = np.ascontiguousarray(text, np.uint8)
    q = np.ascontiguousarray(q, np.uint8)
    if q.size == 0:
        return np.zeros(0, np.uint32)
    cap = 1 << 16
    while True:
        out = np.zeros(cap, np.uint32)
        n = lib().orc_naive_scan(text, text.size, q, q.size, out, cap)
        if n <= cap:
            return out[:n].copy()
        cap = int(n)


def naive_batch(text, qranks, qoff):
    text = np.ascontiguousarray(text, np.uint8)
    qranks = np.ascontiguousarray(qranks, np.uint8)
    qoff = np.ascontiguousarray(qoff, np.uint64)
    nq = qoff.size - 1
    hit_off = np.zeros(nq + 1, np.uint64)
    pos = C.c_void_p()
    qq = qranks if qranks.size else np.zeros(1, np.uint8)
    total = lib().orc_naive_batch(text, text.size, qq, qoff, nq, hit_off, C.byref(pos))
    return hit_off, _take(pos.value, total, np.uint32)
