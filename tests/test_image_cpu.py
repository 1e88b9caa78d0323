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

    cases = {"full_table": (full_table, "differ in number"), "slot_oob": (slot_out_of_region, "outside the element"),
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
