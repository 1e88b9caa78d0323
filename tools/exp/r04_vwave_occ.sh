#!/bin/bash
# Runs ON THE GPU BOX: k_validate_wave compiled for 4 (default) / 5 / 6 waves per SIMD against the repeats probe.
for occ in 4 5 6; do
  echo "== KMX_VWAVE_OCC=$occ"
  KMX_VWAVE_OCC=$occ python -c "from kmer_index_amd import build; build.build(force=True)" || exit 1
  timeout -k 10 300 python tools/probe_skew.py 2>&1 | grep "m=100" | cut -c1-260
done
python -c "from kmer_index_amd import build; build.build(force=True)"
