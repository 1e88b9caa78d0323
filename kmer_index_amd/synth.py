"""Portable synthetic inputs: counter-based SplitMix64 -> multiply-shift ranks.

The reference's generator (benchmarks/input_generator.hpp:52-63) draws i.i.d. uniform ranks
from std::mt19937 + std::uniform_int_distribution<uint8_t>, whose output is implementation
defined; this generator keeps the distribution (i.i.d. uniform over [0, sigma)) and is
identical on every host.  Item i of stream `seed` is rank(mix64(seed + (i+1)*GOLDEN)).
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def mix64(z: np.ndarray) -> np.ndarray:
    z = z.astype(np.uint64, copy=True)
    z ^= z >> np.uint64(30)
    z *= _M1
    z ^= z >> np.uint64(27)
    z *= _M2
    z ^= z >> np.uint64(31)
    return z


def u64_stream(seed: int, n: int, start: int = 0) -> np.ndarray:
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        return mix64(np.uint64(seed) + i * GOLDEN)


def ranks(seed: int, n: int, sigma: int, chunk: int = 1 << 24) -> np.ndarray:
    """n i.i.d. uniform ranks in [0, sigma) as uint8."""
    out = np.empty(n, np.uint8)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        z = u64_stream(seed, m, s)
        out[s:s + m] = (((z >> np.uint64(32)) * np.uint64(sigma)) >> np.uint64(32)).astype(np.uint8)
    return out


def uniform_queries(seed: int, nq: int, m: int, sigma: int):
    """nq queries of length m: (qranks[nq*m] u8, qoff[nq+1] u64)."""
    q = ranks(seed, nq * m, sigma)
    off = np.arange(nq + 1, dtype=np.uint64) * np.uint64(m)
    return q, off


def mixed_queries(seed: int, text: np.ndarray, nq: int, lengths, sigma: int, planted_frac: float = 0.5):
    """Queries with lengths drawn uniformly from `lengths`; a `planted_frac` share is copied from
    the text at a uniform offset (guaranteed hit), the rest are uniform random (cfg 3/5)."""
    lengths = np.asarray(lengths, np.uint64)
    z = u64_stream(seed, nq)
    lens = lengths[((z >> np.uint64(40)) % np.uint64(lengths.size)).astype(np.int64)]
    off = np.zeros(nq + 1, np.uint64)
    np.cumsum(lens, out=off[1:])
    total = int(off[-1])
    q = ranks(seed ^ 0x5DEECE66D, total, sigma)
    z2 = u64_stream(seed + 17, nq)
    planted = (z2 & np.uint64(0xFFFF)).astype(np.float64) < planted_frac * 65536.0
    n = text.size
    idx = np.nonzero(planted)[0]
    if idx.size:
        maxstart = (np.uint64(n) - lens[idx]).astype(np.uint64)
        start = ((z2[idx] >> np.uint64(16)) % (maxstart + np.uint64(1))).astype(np.int64)
        # vectorised ragged copy
        l = lens[idx].astype(np.int64)
        tot = int(l.sum())
        rep = np.repeat(np.arange(idx.size), l)
        within = np.arange(tot) - np.repeat(np.cumsum(l) - l, l)
        dst = off[idx].astype(np.int64)[rep] + within
        src = start[rep] + within
        q[dst] = text[src]
    return q, off
