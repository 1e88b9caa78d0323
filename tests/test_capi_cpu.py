"""CPU suite, part 2: the product library without a GPU — it loads, exports every symbol of include/kmx.h,
its host-only entry points (planner, fast_pow) equal the oracle, and search refuses to run without a device."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "kmx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(kmx_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(engine):
    declared = _declared_symbols()
    assert len(declared) >= 19
    L = engine.lib()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(engine.EXPORTS) == declared
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "kmer_index_amd", "libkmx.so")], capture_output=True, text=True).stdout
    for s in declared:
        assert f" T {s}" in out


def test_library_carries_gfx950_code_objects():
    """The .so embeds a gfx950 code object holding every kernel (no other arch, no fallback)."""
    lib = os.path.join(ROOT, "kmer_index_amd", "libkmx.so")
    data = open(lib, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in data
    for arch in (b"gfx90a", b"gfx942", b"sm_"):
        assert b"amdgcn-amd-amdhsa--" + arch not in data
    for kern in (b"k_lookup", b"k_fill", b"k_validate", b"k_compact", b"k_prefix_merge_pass", b"k_scan_down", b"k_partition"):
        assert kern in data


def test_fast_pow_matches_oracle_and_golden(engine, orc):
    rows = json.load(open(os.path.join(ROOT, "tests", "golden", "fast_pow.json")))["rows"]
    for b, e, want in rows:
        assert engine.fast_pow(b, e) == want == orc.fast_pow(b, e)


@pytest.mark.parametrize("ks", [[5], [10], [8, 10, 12], [9, 11, 13, 17], [3, 4, 5], [31], [12, 8, 10], [1], [2, 9]])
def test_planner_matches_oracle(engine, orc, ks):
    m1, n1 = engine.plan(ks)
    m2, n2 = orc.plan(ks)
    assert np.array_equal(m1, m2)
    assert n1 == n2


def test_planner_thesis_kat(engine):
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "planner.json")))["thesis"]
    multi, nk = engine.plan(gold["ks"])
    assert nk[29] == [9, 9, 11] and nk[30] == [13, 17] and nk[31] == [9, 9, 13] and nk[33] == [9, 11, 13]
    assert not multi[32] and len(nk[32]) == 1


def test_argument_validation_without_device(engine):
    import ctypes as C
    L = engine.lib()
    out = C.c_void_p()
    ranks = np.zeros(100, np.uint8)
    ks = np.array([5], np.uint32)
    # k >= 64 / log2(sigma) is the reference's static_assert (kmer_index.hpp:42-43)
    bad = np.array([32], np.uint32)
    st = L.kmx_index_build(ranks.ctypes.data, 100, 4, bad.ctypes.data, 1, None, C.byref(out))
    assert st == 1 and b"valid k" in L.kmx_last_error()
    st = L.kmx_index_build(ranks.ctypes.data, 3, 4, ks.ctypes.data, 1, None, C.byref(out))
    assert st == 1
    st = L.kmx_index_build(None, 100, 4, ks.ctypes.data, 1, None, C.byref(out))
    assert st == 1


def test_no_cpu_fallback(engine):
    """Without a device the product path fails loudly instead of computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(engine.KmxError) as e:
        engine.Index(np.zeros(1000, np.uint8), 4, [5])
    assert e.value.status == 4      # KMX_ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under kmer_index_amd/ or include/ may reference it."""
    bad = []
    for base in ("kmer_index_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"\boracle\b|liboracle|orc_", txt):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_index_load_rejects_bad_images_before_touching_the_device(engine, tmp_path):
    """kmx_index_load validates magic / sizes / checksum on the host (no GPU needed to refuse a file)."""
    bad = tmp_path / "garbage.kmx"
    bad.write_bytes(b"not an index image at all" * 10)
    with pytest.raises(engine.KmxError) as e:
        engine.Index.load(str(bad))
    assert e.value.status == 1 and "magic" in str(e.value)
    with pytest.raises(engine.KmxError) as e:
        engine.Index.load(str(tmp_path / "missing.kmx"))
    assert e.value.status == 1
    trunc = tmp_path / "trunc.kmx"
    import struct
    trunc.write_bytes(b"KMXIMG01" + struct.pack("<IIQIIIIQ", 2, 4, 1000, 1, 10000, 5, 0, 0))
    with pytest.raises(engine.KmxError) as e:
        engine.Index.load(str(trunc))
    assert e.value.status == 1 and "truncated" in str(e.value)


def test_choose_best_k_matches_oracle_and_known_answer(engine, orc):
    """choose_best_k.hpp:12-60.  Hand-computed: lengths {20, 40, 60} -> 20 gives k=23 one point (miss 3), 40 gives k=21
    two points (miss 2), 60 gives k=21 one point (miss 3): scores 21:3, 23:1, rest 0 in priority order."""
    assert engine.choose_best_k([20, 40, 60], 4) == [21, 23, 29, 27] == orc.choose_best_k([20, 40, 60], 4)
    assert engine.choose_best_k([29 * 3, 58, 27, 10, 11], 3) == orc.choose_best_k([29 * 3, 58, 27, 10, 11], 3)
    rng = np.random.default_rng(3)
    for _ in range(20):
        lens = rng.integers(1, 2000, int(rng.integers(0, 300)))
        n_k = int(rng.integers(1, 11))
        assert engine.choose_best_k(lens, n_k) == orc.choose_best_k(lens, n_k)


def test_header_is_plain_c(tmp_path):
    """include/kmx.h compiles as C99 with -Wall -Wextra -pedantic: the boundary is a C ABI, not a C++ one."""
    obj = tmp_path / "hdr.o"
    res = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{os.path.join(ROOT, 'include')}", "-c",
                          os.path.join(ROOT, "tests", "cpp", "test_header_is_c.c"), "-o", str(obj)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_engine_planner_table_differs_from_the_reference_only_where_it_may(engine):
    """kmx_plan_engine (the table searches run on unless KEEP_MASKS / REFERENCE_PLAN): the reference's planner for every length it
    answers from one k with m <= k and wherever a rest could reach the sub-k fan-out guard (kmer_index.hpp:119-122 via :234) on
    either choice; otherwise the LARGEST k of the index that fits — also for the reference's multi-k sums (re-planned: a sum of
    summands is also a run of parts of ONE k with the k-mer that ends the query as the last of them; 0 = the sum is kept: an exact
    length the scheme owns, :529-530, or every k risky)."""
    rng = np.random.default_rng(12)
    cases = [([8, 10, 12], 4), ([10], 4), ([5], 20), ([9, 10], 4), ([4, 6], 27), ([14, 20, 31], 4), ([3, 4, 5], 15), ([13, 16], 4), ([7, 12], 20)]
    for _ in range(20):
        n_k = int(rng.integers(1, 5))
        sigma = int(rng.choice([2, 4, 5, 15, 20, 27]))
        kmax = {2: 40, 4: 31, 5: 27, 15: 16, 20: 14, 27: 13}[sigma]
        cases.append((sorted(set(int(v) for v in rng.integers(1, kmax + 1, n_k))), sigma))
    limit = 10_000_000
    changed = replanned = 0
    for ks, sigma in cases:
        R = 400
        multi, nk_sum = engine.plan(ks, R)
        used = engine.plan_engine(ks, sigma, R)
        for m in range(1, R):
            risky = lambda k: (m % k) != 0 and sigma ** (k - m % k) > limit      # noqa: E731
            if multi[m] and len(ks) > 1:
                if len(nk_sum[m]) == 1:
                    assert used[m] == 0, (ks, sigma, m)                   # an exact length of a high k: served as an exact lookup
                else:
                    best = max([k for k in ks if k < m and not risky(k)], default=0)
                    assert used[m] == best, (ks, sigma, m, int(used[m]), best)
                    replanned += int(best != 0)
                continue
            k_ref = nk_sum[m][0]
            k_eng = int(used[m])
            assert k_eng in ks
            if m <= k_ref or len(ks) == 1:
                assert k_eng == k_ref, (ks, sigma, m)
                continue
            if risky(k_ref):
                assert k_eng == k_ref, (ks, sigma, m)
                continue
            best = max([k for k in ks if k <= m and k > k_ref and not risky(k)] + [k_ref])
            assert k_eng == best, (ks, sigma, m, k_ref, k_eng, best)
            changed += int(k_eng != k_ref)
    assert changed > 100 and replanned > 100
