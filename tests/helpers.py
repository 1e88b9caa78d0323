"""Shared helpers for the parity tests."""
import numpy as np

from kmer_index_amd import synth


def make_queries(text, sigma, lengths, per_length, seed):
    """Per length: 1/3 uniform random, 1/3 planted at a random offset, 1/3 planted in the text's tail
    (the tail set is what catches a missing last-kmer fix-up, kmer_index.hpp:90-112)."""
    n = text.size
    qs = []
    z = synth.u64_stream(seed, len(lengths) * per_length * 2)
    zi = 0
    for m in lengths:
        for t in range(per_length):
            if m > n or t % 3 == 0:
                q = synth.ranks(seed * 1000003 + m * 131 + t, m, sigma)
            elif t % 3 == 1:
                s = int(z[zi] % np.uint64(n - m + 1))
                q = text[s:s + m].copy()
            else:
                back = int(z[zi] % np.uint64(min(14, n - m) + 1))
                s = n - m - back
                q = text[s:s + m].copy()
            zi += 1
            qs.append(q)
    return pack(qs)


def pack(qs):
    off = np.zeros(len(qs) + 1, np.uint64)
    if qs:
        off[1:] = np.cumsum([len(q) for q in qs])
    ranks = np.concatenate(qs).astype(np.uint8) if qs and off[-1] else np.zeros(0, np.uint8)
    return ranks, off


def digest(hit_off, positions):
    """Order-sensitive 64-bit digest of a whole batch result (FNV-style over counts and positions)."""
    h = np.uint64(1469598103934665603)
    with np.errstate(over="ignore"):
        for arr in (np.diff(hit_off).astype(np.uint64), positions.astype(np.uint64)):
            # chunked polynomial fold: weights are powers of the FNV prime
            for s in range(0, arr.size, 1 << 20):
                a = arr[s:s + (1 << 20)] + np.uint64(1)
                w = np.uint64(1099511628211) ** np.arange(a.size, 0, -1, dtype=np.uint64)
                h = h * (np.uint64(1099511628211) ** np.uint64(a.size)) + np.sum(a * w, dtype=np.uint64)
    return int(h)


def inside_envelope(plan, ks, m):
    """is a query of m letters inside the envelope in which the reference's search() is correct (and the line-by-line
    restatement, MODE_FAITHFUL, therefore equals ground truth)?  plan = orc.plan(ks): (use_multi, nk_sum).
    Single k: any m <= k, exact multiples of k, at most two full parts with a rest (kmer_index.hpp:314 breaks the others);
    multi-k sums: at most two summands (kmer_index.hpp:526, :535)."""
    multi, nk_sum = plan
    if m <= 0 or m >= len(nk_sum) or not nk_sum[m]:
        return True                                   # rejected / unservable lengths: both modes agree on the status
    if multi[m] and len(ks) > 1:
        return len(nk_sum[m]) <= 2
    k = nk_sum[m][0]
    return m <= k or m % k == 0 or m // k <= 2
