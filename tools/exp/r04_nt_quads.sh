#!/bin/bash
# Runs ON THE GPU BOX: the sub-k kernels' quad loads / stores, plain against non-temporal (rebuilds libkmx.so per case, leaves the default
# build behind).  Usage: bash tools/exp/r04_nt_quads.sh > gpurun_out/<dir>/nt_quads.log
for v in KMX_NT_QUAD_LOADS KMX_PLAIN_QUAD_STORES; do
  echo "== $v=1"
  env $v=1 python -c "from kmer_index_amd import build; build.build(force=True)" || exit 1
  timeout -k 10 300 python tools/probe_prefix.py 7 6 3 2>&1 | grep "^m=" | cut -c1-230
done
python -c "from kmer_index_amd import build; build.build(force=True)"
