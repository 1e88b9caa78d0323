#!/bin/bash
# Runs ON THE GPU BOX: positions per value band of the banded slices (KMX_BAND) against the m = 5 probe.
for t in ${KMX_BAND_SWEEP:-6144 4096 5120 7168}; do
  echo "== KMX_BAND=$t"
  KMX_BAND=$t python -c "from kmer_index_amd import build; build.build(force=True)" || exit 1
  timeout -k 10 200 python tools/probe_prefix.py 5 2>&1 | grep "^m=" | cut -c1-330
done
python -c "from kmer_index_amd import build; build.build(force=True)"
