#!/usr/bin/env python3
"""Host model of merge_runs_lds (kmx_kernels.hip): the index arithmetic of the in-LDS merge of R ascending runs —
pair tables, chunk assignment, the gapped layout with one sentinel cell per group, carried groups, the checked first
round — executed thread by thread in lockstep.  Used to validate the scheme before it went to the GPU; kept as
documentation of the layout.  Run: python tools/model_prefix_merge.py"""
import random

SENT = 0xFFFFFFFF


def merge_runs(vals, bnd, NT, EMAX, slack=64):
    """vals: concatenation of R ascending runs (logical layout); bnd: R+1 boundaries.  Returns the sorted list."""
    R = len(bnd) - 1
    n = bnd[R]
    buf = list(vals) + [0xDEADBEEF] * (R + EMAX + slack)       # garbage beyond: reads there must be harmless
    w, ngroups, first = 1, R, True
    # ONE chunk length for every round, from the first round's pair count (later rounds have fewer pairs, so the chunks of a
    # round never outnumber the threads); odd, so that the lanes' places in the buffer spread over the LDS banks
    E = max(3, (-(-n // (NT - (R + 1) // 2))) | 1)
    assert E <= EMAX, (E, EMAX, n, NT, R)
    while ngroups > 1:
        npairs = (ngroups + 1) // 2
        assert npairs < NT
        # pair table (threads 0..npairs-1)
        ptab, c0 = [], 0
        for p in range(npairs):
            g0 = 2 * p
            s, mi, e = bnd[min(g0 * w, R)], bnd[min((g0 + 1) * w, R)], bnd[min((g0 + 2) * w, R)]
            ptab.append((s, mi, e, c0))
            c0 += -(-(e - s) // E)
            if not first and e == mi:
                buf[mi + 2 * p + 1] = SENT                       # a carried group merges with an empty one
        total = c0
        assert total <= NT
        outs = []
        for tid in range(NT):
            if tid >= total:
                continue
            p, step = 0, 1
            while step < npairs:
                step <<= 1
            step >>= 1
            while step:
                c = p + step
                if c < npairs and ptab[c][3] <= tid:
                    p = c
                step >>= 1
            s, mi, e, ch0 = ptab[p]
            na, nb = mi - s, e - mi
            d = (tid - ch0) * E
            nout = min(E, na + nb - d)
            assert nout > 0
            if first:
                pA, pB, a_end, b_end = s, mi, mi, e
            else:
                pA, pB = s + 2 * p, mi + 2 * p + 1
            lo, hi = max(0, d - nb), min(d, na)
            while lo < hi:
                mid = (lo + hi) >> 1
                if buf[pA + mid] < buf[pB + d - 1 - mid]:
                    lo = mid + 1
                else:
                    hi = mid
            pa, pb = pA + lo, pB + d - lo
            if first:
                va = buf[pa] if pa < a_end else SENT
                vb = buf[pb] if pb < b_end else SENT
            else:
                va, vb = buf[pa], buf[pb]
            x = []
            for j in range(E):
                c = va < vb
                x.append(min(va, vb))
                pa += c
                pb += (not c)
                idx = pa if c else pb
                nv = buf[idx]
                if first and idx >= (a_end if c else b_end):
                    nv = SENT
                if c:
                    va = nv
                else:
                    vb = nv
            outs.append((s + p + d, x[:nout]))
        for base, x in outs:                                    # after the barrier
            buf[base:base + len(x)] = x
        for p in range(npairs):                                 # the sentinel cell behind every merged group (thread p)
            buf[ptab[p][2] + p] = SENT
        w *= 2
        ngroups = npairs
        first = False
    return buf[:n]


def main(cases=3000):
    rnd = random.Random(5)
    for case in range(cases):
        NT = rnd.choice([64, 64, 256, 1024])
        R = rnd.randint(1, min(32 if NT == 64 else 64, NT // 2))
        n = rnd.randint(0, rnd.choice([40, 300, 2048]))
        cuts = sorted(rnd.randint(0, n) for _ in range(R - 1))
        bnd = [0] + cuts + [n]
        pool = rnd.sample(range(0, 10 * n + 10), n) if rnd.random() < 0.8 else list(range(n))
        rnd.shuffle(pool)
        vals = []
        for r in range(R):
            vals += sorted(pool[bnd[r]:bnd[r + 1]])
        EMAX = max(3, (-(-max(n, 1) // (NT - (R + 1) // 2))) | 1)
        got = merge_runs(vals, bnd, NT, EMAX)
        assert got == sorted(vals), (case, NT, R, n)
    print("model ok")


if __name__ == "__main__":
    main()
