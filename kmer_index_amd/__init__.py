"""kmer_index_amd — MI355X-native batch k-mer exact-match search behind kmer_index::search().

Only what the hot path needs lives here:
  csrc/      hand-written gfx950 kernels + the C-ABI (include/kmx.h)
  engine.py  ctypes binding of the C-ABI (plumbing for tests / bench / smoke)
  synth.py   portable synthetic inputs
  build.py   hipcc build of libkmx.so
The C++ host mirror of the reference's template surface is in include/kmer_index_amd/.
"""
from . import build, engine, synth  # noqa: F401
