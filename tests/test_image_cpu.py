"""kmx_index_load's content validation: an image whose size fields and checksum are right but whose tables
are wrong is refused on the host, before any device is touched — a full slot table would make the probe loop spin on
the GPU, an offset past the region would read out of bounds."""
import copy

import numpy as np
import pytest

from kmer_index_amd import synth
from tests import image_writer as iw


def _elems():
    text = synth.ranks(77, 3000, 4)
    return text, [iw.flatten(text, 4, 4, 2), iw.flatten(text, 4, 7, 1)]


def _load_error(engine, tmp_path, text, elems, name):
    p = tmp_path / f"{name}.kmx"
    iw.write_image(str(p), text, 4, elems)
    with pytest.raises(engine.KmxError) as e:
        engine.Index.load(str(p))
    return e.value


def test_valid_third_party_image_passes_validation(engine, tmp_path):
    """The numpy-written image gets through every host-side check; without a GPU the next step refuses (no device)."""
    import torch
    text, elems = _elems()
    p = tmp_path / "ok.kmx"
    iw.write_image(str(p), text, 4, elems)
    if torch.cuda.is_available():
        idx = engine.Index.load(str(p))
        assert idx.info()["ks"] == [4, 7]
        idx.close()
    else:
        with pytest.raises(engine.KmxError) as e:
            engine.Index.load(str(p))
        assert e.value.status == 4, str(e.value)          # KMX_ERR_NO_DEVICE: everything before the device passed


def test_corrupt_contents_are_refused_on_the_host(engine, tmp_path):
    text, good = _elems()

    def variant(fn):
        el = copy.deepcopy(good)
        fn(el)
        return el

    def full_table(el):                                     # every slot occupied: probe() of a missing key never ends
        s = el[1]["slots"]
        s["cnt"][s["cnt"] == 0] = 1

    def slot_out_of_region(el):
        s = el[1]["slots"]
        i = int(np.nonzero(s["cnt"])[0][0])
        s["off"][i] = el[1]["region"] - 1
        s["cnt"][i] = 5

    def unsorted_keys(el):
        el[1]["ukeys"][[3, 4]] = el[1]["ukeys"][[4, 3]]

    def key_outside_key_space(el):
        el[1]["ukeys"][-1] = el[1]["n_keys"] + 5

    def offs_not_monotone(el):
        el[0]["offs"][10] = el[0]["offs"][11] + 7

    def offs_wrong_end(el):
        el[0]["offs"][-1] -= 1

    def position_outside_text(el):
        el[0]["positions"][5] = text.size + 100

    def group_not_sorted(el):
        o = el[0]["offs"]
        j = int(np.nonzero(np.diff(o) >= 2)[0][0])
        a = int(o[j])
        el[0]["positions"][[a, a + 1]] = el[0]["positions"][[a + 1, a]]

    def slot_names_half_a_group(el):
        s = el[1]["slots"]
        i = int(np.nonzero(s["cnt"] >= 2)[0][0])
        s["cnt"][i] -= 1

    cases = {"full_table": (full_table, "differ in number"), "unsorted_group": (group_not_sorted, "strictly ascending"),
             "slot_half_group": (slot_names_half_a_group, "does not name the group"), "slot_oob": (slot_out_of_region, "outside the element"),
             "unsorted_keys": (unsorted_keys, "ascending"), "key_range": (key_outside_key_space, "ascending"),
             "offs_monotone": (offs_not_monotone, "monotone"), "offs_end": (offs_wrong_end, "span"),
             "position_range": (position_outside_text, "outside the text")}
    for name, (fn, needle) in cases.items():
        err = _load_error(engine, tmp_path, text, variant(fn), name)
        assert err.status == 1 and "corrupt contents" in str(err) and needle in str(err), (name, str(err))


def test_options_struct_of_version_1_is_still_accepted(engine):
    """A caller compiled against KMX_VERSION 1 passes the shorter kmx_options: accepted (the call proceeds to the next check),
    any other size is refused."""
    import ctypes as C
    L = engine.lib()
    out = C.c_void_p()
    ranks = np.zeros(100, np.uint8)
    ks = np.array([5], np.uint32)
    o = engine.Options()
    o.struct_size = 32
    o.device = -1
    st = L.kmx_index_build(None, 100, 4, ks.ctypes.data, 1, C.byref(o), C.byref(out))
    assert st == 1 and b"struct_size" not in L.kmx_last_error()
    o.struct_size = 36
    st = L.kmx_index_build(ranks.ctypes.data, 100, 4, ks.ctypes.data, 1, C.byref(o), C.byref(out))
    assert st == 1 and b"struct_size" in L.kmx_last_error()


# ---------------------------------------------------------------------------------------------------------------------
# mutation fuzz of the loader: whatever bytes the file holds, kmx_index_load returns a status — it never crashes, never
# allocates on the word of a header the file cannot back, and (GPU leg) an image it ACCEPTS can be searched without harm.
# ---------------------------------------------------------------------------------------------------------------------
_EXTREME = [0, 1, 2, 31, 32, 63, 64, 255, 256, 65535, 65536, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFE, 0xFFFFFFFF,
            1 << 32, (1 << 40) + 1, 1 << 62, (1 << 63) - 1, 1 << 63, (1 << 64) - 1]


def _sections(data, n_ks, sizes):
    """(offset, length) of every checksummed section of an image laid out with the ORIGINAL sizes."""
    out, at = [], 48
    for ln in sizes:
        out.append((at, ln))
        at += (ln + 7) // 8 * 8
    return out


def _mutants(rng, text, elems, n):
    """n byte strings: the valid image with a few bytes / fields changed, the checksum made right again."""
    import io
    import os
    import tempfile
    fd, path = tempfile.mkstemp(suffix=".kmx")
    os.close(fd)
    iw.write_image(path, text, 4, elems)
    good = open(path, "rb").read()
    os.remove(path)
    kmax = max(e["k"] for e in elems)
    sizes = [72 * len(elems), kmax]
    for e in elems:
        sizes += [4 * e["positions"].size, 4 * e["offs"].size, 4 * e["atab"].size, 16 * e["slots"].size, 8 * e["ukeys"].size]
    secs = _sections(good, len(elems), sizes)
    head_end = secs[1][0] + 8                                 # header + element table + tail
    for _ in range(n):
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            how = int(rng.integers(0, 5))
            if how == 0:                                      # a 32-bit field of the header / element table
                at = int(rng.integers(2, head_end // 4)) * 4
                b[at:at + 4] = int(_EXTREME[int(rng.integers(0, len(_EXTREME)))] & 0xFFFFFFFF).to_bytes(4, "little")
            elif how == 1:                                    # a 64-bit field of the element table
                at = 48 + int(rng.integers(0, (head_end - 48) // 8)) * 8
                b[at:at + 8] = int(_EXTREME[int(rng.integers(0, len(_EXTREME)))]).to_bytes(8, "little")
            elif how == 2:                                    # a 32-bit word anywhere (positions, offsets, slots, keys)
                at = int(rng.integers(12, len(b) // 4)) * 4
                b[at:at + 4] = int(_EXTREME[int(rng.integers(0, len(_EXTREME)))] & 0xFFFFFFFF).to_bytes(4, "little")
            elif how == 3:                                    # one bit anywhere
                at = int(rng.integers(8, len(b)))
                b[at] ^= 1 << int(rng.integers(0, 8))
            else:                                             # a small change of a small field (k, table kind, counts +- 1)
                at = 48 + int(rng.integers(0, (head_end - 48) // 4)) * 4
                v = (int.from_bytes(b[at:at + 4], "little") + int(rng.integers(-2, 3))) & 0xFFFFFFFF
                b[at:at + 4] = v.to_bytes(4, "little")
        mx = iw.Mixer()
        for at, ln in secs:
            mx.add(bytes(b[at:at + ln]))
        b[40:48] = mx.h.to_bytes(8, "little")
        cut = int(rng.integers(0, 8))
        if cut == 0:
            b = b[:int(rng.integers(0, len(b)))]
        elif cut == 1:
            b += bytes(int(rng.integers(1, 64)))
        yield bytes(b)


def test_loader_survives_mutated_images(engine, tmp_path):
    import torch
    rng = np.random.default_rng(20260)
    text = synth.ranks(78, 700, 4)
    elems = [iw.flatten(text, 4, 3, 2), iw.flatten(text, 4, 6, 1)]
    p = tmp_path / "m.kmx"
    seen = {}
    for raw in _mutants(rng, text, elems, 400):
        p.write_bytes(raw)
        try:
            idx = engine.Index.load(str(p))
        except engine.KmxError as e:
            seen[e.status] = seen.get(e.status, 0) + 1
            continue
        assert torch.cuda.is_available()                      # only a box with a GPU can get past the device check
        idx.close()
        seen[0] = seen.get(0, 0) + 1
    assert seen.get(1, 0) > 100, seen                         # most mutants are refused as invalid
    assert set(seen) <= {0, 1, 4}, seen                       # INVALID_ARGUMENT, or valid-but-no-device; never out-of-memory
