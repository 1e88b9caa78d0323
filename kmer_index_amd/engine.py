"""ctypes binding of the C-ABI (include/kmx.h) — plumbing for tests, bench.py and smoke().

Every search goes through libkmx.so (hand-written gfx950 kernels).  There is no Python or
CPU search path in this package: when the library or a device is missing, calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

KMX_MAX_KS = 32
KMX_MAX_DEVICES = 16
KMX_N_KERNELS = 16
TABLE_AUTO, TABLE_OPEN, TABLE_DENSE = 0, 1, 2
SEARCH_DEFAULT, SEARCH_KEEP_MASKS, SEARCH_COUNT_ONLY, SEARCH_ASYNC, SEARCH_REFERENCE_PLAN = 0, 1, 2, 4, 8
KIND_NONE, KIND_EXACT, KIND_STITCH, KIND_PREFIX = 0, 1, 2, 3
Q_OK, Q_TOO_LONG, Q_SUBK_FANOUT, Q_EMPTY_QUERY, Q_BAD_RANK = 0, 1, 2, 3, 4

# every symbol include/kmx.h declares
EXPORTS = [
    "kmx_index_build", "kmx_index_free", "kmx_index_save", "kmx_index_load", "kmx_index_info", "kmx_index_memory", "kmx_index_arena_host", "kmx_index_extend_query_size_range", "kmx_choose_best_k", "kmx_plan", "kmx_plan_engine", "kmx_fast_pow",
    "kmx_search_batch", "kmx_search_batch_device", "kmx_result_counts", "kmx_result_view_device",
    "kmx_result_view", "kmx_result_masks", "kmx_result_free", "kmx_stats_enable", "kmx_stats_get",
    "kmx_stats_reset", "kmx_debug_words", "kmx_last_error", "kmx_status_string", "kmx_version",
    "kmx_index_devices", "kmx_result_parts", "kmx_result_part_view_device",
    "kmx_index_bucket_host", "kmx_index_levels", "kmx_result_gather_device",
]


class KmxError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"kmx status {status}: {msg}")
        self.status = status


class Options(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("table_kind", C.c_uint32),
                ("n_threads", C.c_uint32), ("query_size_range", C.c_uint32), ("keep_host_arena", C.c_uint32),
                ("host_flatten", C.c_uint32), ("no_aligned_copy", C.c_uint32),
                ("n_devices", C.c_uint32), ("devices", C.c_int32 * KMX_MAX_DEVICES), ("prefix_levels", C.c_int32)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char_p), ("launches", C.c_uint64), ("total_ms", C.c_double)]


_lib = None


def lib():
    """Loads kmer_index_amd/libkmx.so, building it first when stale (hipcc, gfx950)."""
    global _lib
    if _lib is None:
        # One HIP runtime per process: torch ships its own libamdhip64.so.7 and finds no GPU when another copy
        # (the /opt/rocm one libkmx.so links to) was mapped first.  Importing torch first makes both share torch's.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        path = _build.build()
        if not os.path.exists(path):
            raise RuntimeError("libkmx.so is missing and could not be built; the engine has no fallback")
        L = C.CDLL(path)
        vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
        P = C.POINTER
        L.kmx_index_build.restype = C.c_int
        L.kmx_index_build.argtypes = [vp, u64, u32, vp, u32, P(Options), P(vp)]
        L.kmx_index_free.argtypes = [vp]
        L.kmx_index_save.restype = C.c_int
        L.kmx_index_save.argtypes = [vp, C.c_char_p]
        L.kmx_index_load.restype = C.c_int
        L.kmx_index_load.argtypes = [C.c_char_p, P(Options), P(vp)]
        L.kmx_index_info.restype = C.c_int
        L.kmx_index_info.argtypes = [vp, P(u64), P(u32), P(u32), vp, vp, P(u64)]
        L.kmx_index_memory.restype = C.c_int
        L.kmx_index_memory.argtypes = [vp, P(u64), P(u64), P(u64), P(u64), P(u64)]
        L.kmx_index_arena_host.restype = C.c_int
        L.kmx_index_arena_host.argtypes = [vp, P(vp), P(u64)]
        L.kmx_index_extend_query_size_range.restype = C.c_int
        L.kmx_index_extend_query_size_range.argtypes = [vp, u32]
        L.kmx_choose_best_k.restype = C.c_int
        L.kmx_choose_best_k.argtypes = [vp, u64, u32, vp]
        L.kmx_plan.restype = C.c_int
        L.kmx_plan_engine.restype = C.c_int
        L.kmx_plan_engine.argtypes = [vp, u32, u32, u32, vp]
        L.kmx_plan.argtypes = [vp, u32, u32, vp, vp, vp, u64, P(u64)]
        L.kmx_fast_pow.restype = u64
        L.kmx_fast_pow.argtypes = [u64, C.c_uint8]
        L.kmx_search_batch.restype = C.c_int
        L.kmx_search_batch.argtypes = [vp, vp, vp, u64, u32, P(vp)]
        L.kmx_search_batch_device.restype = C.c_int
        L.kmx_search_batch_device.argtypes = [vp, vp, vp, u64, u32, vp, P(vp)]
        L.kmx_result_counts.restype = C.c_int
        L.kmx_result_counts.argtypes = [vp] + [P(u64)] * 6
        L.kmx_result_view_device.restype = C.c_int
        L.kmx_result_view_device.argtypes = [vp, P(vp), P(vp), P(vp)]
        L.kmx_result_view.restype = C.c_int
        L.kmx_result_view.argtypes = [vp, P(vp), P(vp), P(vp), P(vp)]
        L.kmx_result_masks.restype = C.c_int
        L.kmx_result_masks.argtypes = [vp, P(vp), P(vp), P(vp), P(vp)]
        L.kmx_result_free.argtypes = [vp]
        L.kmx_index_devices.restype = C.c_int
        L.kmx_index_devices.argtypes = [vp, P(u32), vp]
        L.kmx_result_parts.restype = C.c_int
        L.kmx_result_parts.argtypes = [vp, P(u32)]
        L.kmx_result_part_view_device.restype = C.c_int
        L.kmx_result_part_view_device.argtypes = [vp, u32, P(C.c_int32), P(u64), P(u64), P(vp), P(vp), P(vp)]
        L.kmx_index_bucket_host.restype = C.c_int
        L.kmx_index_bucket_host.argtypes = [vp, u32, vp, P(vp), P(u32)]
        L.kmx_index_levels.restype = C.c_int
        L.kmx_index_levels.argtypes = [vp, vp]
        L.kmx_result_gather_device.restype = C.c_int
        L.kmx_result_gather_device.argtypes = [vp, C.c_int32, P(vp), P(vp), P(vp)]
        L.kmx_stats_enable.restype = C.c_int
        L.kmx_stats_enable.argtypes = [vp, C.c_int]
        L.kmx_stats_get.restype = C.c_int
        L.kmx_stats_get.argtypes = [vp, P(KernelStat), P(u32)]
        L.kmx_stats_reset.restype = C.c_int
        L.kmx_stats_reset.argtypes = [vp]
        L.kmx_debug_words.restype = C.c_int
        L.kmx_debug_words.argtypes = [vp, vp]
        L.kmx_last_error.restype = C.c_char_p
        L.kmx_status_string.restype = C.c_char_p
        L.kmx_status_string.argtypes = [C.c_int]
        L.kmx_version.restype = u32
        _lib = L
    return _lib


def _set_devices(o, devices):
    if devices is None:
        o.n_devices = 0
        return
    devices = list(devices)
    if not 1 <= len(devices) <= KMX_MAX_DEVICES:
        raise ValueError("devices: between 1 and KMX_MAX_DEVICES ordinals")
    o.n_devices = len(devices)
    for i, d in enumerate(devices):
        o.devices[i] = int(d)
    o.device = int(devices[0])


def _check(st):
    if st != 0:
        raise KmxError(st, lib().kmx_last_error().decode())


def fast_pow(base, exp):
    return int(lib().kmx_fast_pow(base, exp))


def choose_best_k(lengths, n_k=4):
    """kmx_choose_best_k (choose_best_k.hpp): recommended ks for a set of query lengths."""
    lengths = np.ascontiguousarray(lengths, np.uint64)
    out = np.zeros(n_k, np.uint32)
    _check(lib().kmx_choose_best_k(lengths.ctypes.data, lengths.size, n_k, out.ctypes.data))
    return out.tolist()


def plan(ks, rng=10000):
    """(use_multi[rng] bool, nk_sum list of lists) from kmx_plan (host only, no device needed)."""
    ks = np.ascontiguousarray(ks, np.uint32)
    multi = np.zeros(rng, np.uint8)
    off = np.zeros(rng + 1, np.uint32)
    n = C.c_uint64()
    _check(lib().kmx_plan(ks.ctypes.data, ks.size, rng, multi.ctypes.data, off.ctypes.data, None, 0, C.byref(n)))
    flat = np.zeros(max(n.value, 1), np.uint32)
    _check(lib().kmx_plan(ks.ctypes.data, ks.size, rng, multi.ctypes.data, off.ctypes.data, flat.ctypes.data, n.value, C.byref(n)))
    return multi.astype(bool), [flat[off[q]:off[q + 1]].tolist() for q in range(rng)]


def plan_engine(ks, sigma, rng=10000):
    """k_used[rng] from kmx_plan_engine: the k whose element the ENGINE answers a single-k length from (0: multi-k scheme)."""
    ks = np.ascontiguousarray(ks, np.uint32)
    out = np.zeros(rng, np.uint32)
    _check(lib().kmx_plan_engine(ks.ctypes.data, ks.size, rng, int(sigma), out.ctypes.data))
    return out


def _view(ptr, n, dtype):
    if not ptr or n == 0:
        return np.zeros(0, dtype)
    ct = np.ctypeslib.as_ctypes_type(dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(int(n),))


class Result:
    """Owns a kmx_result handle."""

    def __init__(self):
        self._h = C.c_void_p()
        self._index = None        # the Index of the last search: a pending (SEARCH_ASYNC) search reads it when it completes

    def n_parts(self):
        n = C.c_uint32()
        _check(lib().kmx_result_parts(self._h, C.byref(n)))
        return int(n.value)

    def part_device_ptrs(self, part):
        """(device ordinal, q_begin, q_end, d_hit_off, d_positions, d_status) of one part of a multi-device result;
        hit_off is local to the part."""
        dev, qb, qe = C.c_int32(), C.c_uint64(), C.c_uint64()
        a, b, s = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().kmx_result_part_view_device(self._h, part, C.byref(dev), C.byref(qb), C.byref(qe), C.byref(a), C.byref(b), C.byref(s)))
        return int(dev.value), int(qb.value), int(qe.value), a.value, b.value, s.value

    def counts(self):
        v = [C.c_uint64() for _ in range(6)]
        _check(lib().kmx_result_counts(self._h, *[C.byref(x) for x in v]))
        return dict(zip(["nq", "n_hits", "n_exact", "n_stitch", "n_prefix", "n_error"], [int(x.value) for x in v]))

    def host(self, copy=True):
        """(hit_off[nq+1], positions, status[nq], kinds[nq]) as numpy arrays: copies, or with copy=False views of the
        result's own host buffers (valid until the result is searched into again or closed)."""
        c = self.counts()
        a, b, s, k = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().kmx_result_view(self._h, C.byref(a), C.byref(b), C.byref(s), C.byref(k)))
        nq = c["nq"]
        hit_off = _view(a.value, nq + 1, np.uint64)
        n_pos = int(hit_off[nq]) if b.value else 0        # positions are NULL for COUNT_ONLY results
        out = (hit_off, _view(b.value, n_pos, np.uint32), _view(s.value, nq, np.uint8), _view(k.value, nq, np.uint8))
        return tuple(x.copy() for x in out) if copy else out

    def device_ptrs(self):
        a, b, s = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().kmx_result_view_device(self._h, C.byref(a), C.byref(b), C.byref(s)))
        return a.value, b.value, s.value

    def device_tensors(self, device):
        """(hit_off int64 [nq+1], positions int32 [n_hits]) as torch tensors aliasing the result's HBM buffers."""
        import torch

        class _Arr:
            def __init__(self, ptr, n, typestr):
                self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}

        c = self.counts()
        a, b, _ = self.device_ptrs()
        t_off = torch.as_tensor(_Arr(a, c["nq"] + 1, "<i8"), device=device)
        t_pos = torch.as_tensor(_Arr(b, c["n_hits"], "<i4"), device=device) if c["n_hits"] else torch.empty(0, dtype=torch.int32, device=device)
        return t_off, t_pos

    def gather_device(self, dst_device):
        """kmx_result_gather_device: (d_hit_off, d_positions, d_status) of the whole batch in the HBM of dst_device."""
        a, b, s = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().kmx_result_gather_device(self._h, int(dst_device), C.byref(a), C.byref(b), C.byref(s)))
        return a.value, b.value, s.value

    def masks(self):
        c = self.counts()
        a, b, cc, d = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().kmx_result_masks(self._h, C.byref(a), C.byref(b), C.byref(cc), C.byref(d)))
        nq = c["nq"]
        base = _view(a.value, nq, np.uint64).copy()
        cnt = _view(cc.value, nq, np.uint32).copy()
        src = _view(d.value, nq, np.uint64).copy()
        return base, b.value, cnt, src

    def close(self):
        if self._h:
            lib().kmx_result_free(self._h)
            self._h = C.c_void_p()
        self._index = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Index:
    """kmx_index handle: the flattened kmer_index<alphabet_t, uint32_t, ks...> resident in HBM."""

    def __init__(self, ranks, sigma, ks, table=TABLE_AUTO, device=-1, n_threads=0, keep_host_arena=False,
                 query_size_range=0, host_flatten=False, aligned_copy=True, devices=None, prefix_levels=0):
        """devices: None = one replica on `device` (KMX_DEVICES in the environment may widen it); a list of ordinals =
        built on devices[0] and replicated onto the others (host-buffer searches then shard over the replicas).
        prefix_levels: kmx_options.prefix_levels (0 = default, -1 = none, N = at most N pre-merged levels per element)."""
        ranks = np.ascontiguousarray(ranks, np.uint8)
        ks = np.ascontiguousarray(ks, np.uint32)
        self.ks = ks.tolist()
        self.sigma = int(sigma)
        self.n = int(ranks.size)
        o = Options()
        o.struct_size = C.sizeof(Options)
        o.device = device
        o.table_kind = table
        o.n_threads = n_threads
        o.query_size_range = query_size_range
        o.keep_host_arena = int(keep_host_arena)
        o.host_flatten = int(host_flatten)
        o.no_aligned_copy = int(not aligned_copy)
        o.prefix_levels = int(prefix_levels)
        _set_devices(o, devices)
        self._h = C.c_void_p()
        _check(lib().kmx_index_build(ranks.ctypes.data, ranks.size, self.sigma, ks.ctypes.data, ks.size,
                                     C.byref(o), C.byref(self._h)))

    @classmethod
    def load(cls, path, device=-1, keep_host_arena=False, devices=None, prefix_levels=0):
        """kmx_index_load: an index from an image written by save()."""
        self = cls.__new__(cls)
        o = Options()
        o.struct_size = C.sizeof(Options)
        o.device = device
        o.keep_host_arena = int(keep_host_arena)
        o.prefix_levels = int(prefix_levels)
        _set_devices(o, devices)
        self._h = C.c_void_p()
        _check(lib().kmx_index_load(os.fsencode(path), C.byref(o), C.byref(self._h)))
        info = self.info()
        self.ks, self.sigma, self.n = info["ks"], info["sigma"], info["n"]
        return self

    def save(self, path):
        _check(lib().kmx_index_save(self._h, os.fsencode(path)))

    def info(self):
        n, sigma, nks, dbytes = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        ks = np.zeros(KMX_MAX_KS, np.uint32)
        tk = np.zeros(KMX_MAX_KS, np.uint32)
        _check(lib().kmx_index_info(self._h, C.byref(n), C.byref(sigma), C.byref(nks), ks.ctypes.data, tk.ctypes.data, C.byref(dbytes)))
        return {"n": n.value, "sigma": sigma.value, "ks": ks[:nks.value].tolist(), "tables": tk[:nks.value].tolist(),
                "device_bytes": dbytes.value}

    def memory(self):
        """kmx_index_memory: device bytes of one replica by part (positions, aligned_copy, cells, prefix_levels, tables)."""
        v = [C.c_uint64() for _ in range(5)]
        _check(lib().kmx_index_memory(self._h, *[C.byref(x) for x in v]))
        return dict(zip(["positions", "aligned_copy", "cells", "prefix_levels", "tables"], [int(x.value) for x in v]))

    def devices(self):
        n = C.c_uint32()
        d = (C.c_int32 * KMX_MAX_DEVICES)()
        _check(lib().kmx_index_devices(self._h, C.byref(n), d))
        return [int(d[i]) for i in range(n.value)]

    def levels(self):
        """kmx_index_levels: prefix levels built per element (info()["ks"] order)."""
        lv = np.zeros(KMX_MAX_KS, np.uint32)
        _check(lib().kmx_index_levels(self._h, lv.ctypes.data))
        return lv[:len(self.ks)].tolist()

    def bucket_host(self, k, ranks):
        """kmx_index_bucket_host: the bucket of one k-mer out of the host arena (search_k, kmer_index.hpp:183-190); None on a miss."""
        ranks = np.ascontiguousarray(ranks, np.uint8)
        if ranks.size != k:
            raise ValueError("bucket_host: k letters expected")
        p, n = C.c_void_p(), C.c_uint32()
        _check(lib().kmx_index_bucket_host(self._h, int(k), ranks.ctypes.data, C.byref(p), C.byref(n)))
        return _view(p.value, n.value, np.uint32).copy() if p.value else None

    def extend_query_size_range(self, new_maximum):
        _check(lib().kmx_index_extend_query_size_range(self._h, new_maximum))

    def arena_host(self):
        p, n = C.c_void_p(), C.c_uint64()
        _check(lib().kmx_index_arena_host(self._h, C.byref(p), C.byref(n)))
        return _view(p.value, n.value, np.uint32)

    def search(self, qranks, qoff, flags=SEARCH_DEFAULT, result=None):
        """Host-buffer batch search (kmx_search_batch)."""
        qranks = np.ascontiguousarray(qranks, np.uint8)
        qoff = np.ascontiguousarray(qoff, np.uint64)
        r = result or Result()
        _check(lib().kmx_search_batch(self._h, qranks.ctypes.data if qranks.size else None, qoff.ctypes.data,
                                      qoff.size - 1, flags, C.byref(r._h)))
        r._index = self
        return r

    def search_device(self, d_qranks_ptr, d_qoff_ptr, nq, flags=SEARCH_DEFAULT, stream=0, result=None):
        """Device-buffer batch search (kmx_search_batch_device) on a caller-owned hipStream_t."""
        r = result or Result()
        _check(lib().kmx_search_batch_device(self._h, d_qranks_ptr, d_qoff_ptr, nq, flags, stream or None, C.byref(r._h)))
        r._index = self
        return r

    def debug_words(self):
        w = np.zeros(16, np.uint64)
        _check(lib().kmx_debug_words(self._h, w.ctypes.data))
        return w

    def stats_enable(self, on=True):
        _check(lib().kmx_stats_enable(self._h, int(on)))

    def stats_reset(self):
        _check(lib().kmx_stats_reset(self._h))

    def stats(self):
        arr = (KernelStat * KMX_N_KERNELS)()
        n = C.c_uint32()
        _check(lib().kmx_stats_get(self._h, arr, C.byref(n)))
        return {arr[i].name.decode(): {"launches": int(arr[i].launches), "total_ms": float(arr[i].total_ms)} for i in range(n.value)}

    def close(self):
        if self._h:
            lib().kmx_index_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def split_hits(hit_off, positions):
    """List of per-query position arrays."""
    return [positions[int(hit_off[i]):int(hit_off[i + 1])] for i in range(len(hit_off) - 1)]
