"""An index image that kmx_index_load ACCEPTS can be searched without harm, whatever it holds: the loader's content checks
are what stands between a file and the kernels' indexing (a full slot table would spin the probe loop, an offset past the
region would read out of bounds).  Mutants of a valid image — fields and words changed, checksum made right again — are
loaded; every accepted one runs a batch of queries of all kinds (exact, sub-k prefix, stitched, too long, empty).  The
answers of a wrong image are wrong; what is asserted is that every call returns and the result is well formed."""
import numpy as np
import pytest

from kmer_index_amd import synth
from tests import image_writer as iw
from tests.test_image_cpu import _mutants

pytestmark = pytest.mark.gpu


def _queries(rng, text, n):
    lens = rng.integers(0, 26, n)
    parts, off = [], [0]
    for ln in lens.tolist():
        if ln and rng.random() < 0.6:
            s = int(rng.integers(0, text.size - ln + 1))
            parts.append(text[s:s + ln])
        else:
            parts.append(rng.integers(0, 4, ln).astype(np.uint8))
        off.append(off[-1] + ln)
    return (np.concatenate(parts) if parts else np.zeros(0, np.uint8)), np.array(off, np.uint64)


def test_accepted_mutants_are_searchable(engine, tmp_path):
    rng = np.random.default_rng(4711)
    text = synth.ranks(78, 700, 4)
    elems = [iw.flatten(text, 4, 3, 2), iw.flatten(text, 4, 6, 1)]
    qr, qoff = _queries(rng, text, 3000)
    p = tmp_path / "m.kmx"
    accepted = refused = 0
    for raw in _mutants(rng, text, elems, 600):
        p.write_bytes(raw)
        try:
            idx = engine.Index.load(str(p))
        except engine.KmxError as e:
            assert e.status == 1, str(e)
            refused += 1
            continue
        accepted += 1
        for sl in (slice(0, 3000), slice(0, 40), slice(7, 8)):           # the batch pipeline and the latency path
            q0, q1 = sl.start, sl.stop
            r = idx.search(qr[int(qoff[q0]):int(qoff[q1])], qoff[q0:q1 + 1] - qoff[q0])
            hit_off, pos, status, kinds = r.host()
            assert hit_off[0] == 0 and np.all(np.diff(hit_off.astype(np.int64)) >= 0) and int(hit_off[-1]) == pos.size
            assert status.size == q1 - q0
            r.close()
        idx.close()
    assert accepted >= 30 and refused >= 300, (accepted, refused)
