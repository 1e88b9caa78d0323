"""An independent writer of the on-disk index image (the format documented above kmx_index_save in kmx_capi.hip),
in numpy — test infrastructure: lets the CPU suite hand kmx_index_load images that are VALID in every size field and
checksum but wrong in their contents, and lets the GPU suite check that a third-party image loads and searches."""
import struct

import numpy as np

MASK = (1 << 64) - 1
GOLD = 0x9E3779B97F4A7C15


class Mixer:
    def __init__(self):
        self.h = GOLD

    def add(self, b: bytes):
        n8 = len(b) // 8 * 8
        h = self.h
        for (w,) in struct.iter_unpack("<Q", b[:n8]):
            h = ((h ^ w) * 0x100000001B3) & MASK
            h ^= h >> 29
        if n8 < len(b):
            w = int.from_bytes(b[n8:], "little")
            h = ((h ^ w) * 0x100000001B3) & MASK
            h ^= h >> 29
        self.h = h


def _pad(b: bytes) -> bytes:
    return b + b"\0" * ((8 - len(b) % 8) % 8)


def flatten(text, sigma, k, table):
    """One element: positions grouped by rank-hash (ascending inside a group), offs, [ukeys, slots]."""
    n = text.size
    npos = n - k + 1
    h = np.zeros(npos, np.uint64)
    for j in range(k):
        h = h * np.uint64(sigma) + text[j:j + npos].astype(np.uint64)
    order = np.argsort(h, kind="stable").astype(np.uint32)
    hs = h[order]
    n_keys = sigma ** k
    el = {"k": k, "table": table, "n_keys": n_keys, "npos": npos, "positions": order, "region": npos, "atab": np.zeros(0, np.uint32)}
    if table == 2:   # dense
        el["offs"] = np.searchsorted(hs, np.arange(n_keys + 1, dtype=np.uint64)).astype(np.uint32)
        el["ukeys"] = np.zeros(0, np.uint64)
        el["slots"] = np.zeros(0, np.dtype([("key", "<u8"), ("off", "<u4"), ("cnt", "<u4")]))
        el["log2cap"] = 0
    else:
        uk, first = np.unique(hs, return_index=True)
        offs = np.append(first, npos).astype(np.uint32)
        log2cap = max(4, int(np.ceil(np.log2(2 * max(uk.size, 1)))))
        cap = 1 << log2cap
        slots = np.zeros(cap, np.dtype([("key", "<u8"), ("off", "<u4"), ("cnt", "<u4")]))
        for i, key in enumerate(uk.tolist()):
            s = ((key * GOLD) & MASK) >> (64 - log2cap)
            while slots["cnt"][s]:
                s = (s + 1) & (cap - 1)
            slots[s] = (key, offs[i], offs[i + 1] - offs[i])
        el.update(offs=offs, ukeys=uk.astype(np.uint64), slots=slots, log2cap=log2cap)
    return el


def write_image(path, text, sigma, elems, query_range=10000):
    kmax = max(e["k"] for e in elems)
    mx = Mixer()
    body = b""
    fes = b""
    for e in elems:
        fes += struct.pack("<IIIIQQQQQQQ", e["k"], e["table"], e["log2cap"], 0, e["n_keys"], e["npos"], e["offs"].size,
                           e["slots"].size, e["ukeys"].size, e["region"], e["atab"].size)
    sections = [fes, text[text.size - kmax:].astype(np.uint8).tobytes()]
    for e in elems:
        sections += [e["positions"].astype("<u4").tobytes(), e["offs"].astype("<u4").tobytes(), e["atab"].astype("<u4").tobytes(),
                     e["slots"].tobytes(), e["ukeys"].astype("<u8").tobytes()]
    for sec in sections:
        mx.add(sec)
        body += _pad(sec)
    hdr = b"KMXIMG01" + struct.pack("<IIQIIIIQ", 2, sigma, text.size, len(elems), query_range, kmax, 0, mx.h)
    with open(path, "wb") as f:
        f.write(hdr + body)
