#!/bin/bash
# Runs ON THE GPU BOX: threads / waves per SIMD of the 256-thread sub-k shape (k_prefix_sort_block, k_prefix_merge_band) against m = 7 and m = 5.
for cfg in "256 4" "512 8" "512 6" "256 5"; do
  set -- $cfg
  echo "== KMX_MID_THREADS=$1 KMX_MID_OCC=$2"
  KMX_MID_THREADS=$1 KMX_MID_OCC=$2 python -c "from kmer_index_amd import build; build.build(force=True)" || exit 1
  timeout -k 10 200 python tools/probe_prefix.py 7 5 2>&1 | grep "^m=" | cut -c1-330
done
python -c "from kmer_index_amd import build; build.build(force=True)"
