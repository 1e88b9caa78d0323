#!/bin/bash
# Runs ON THE GPU BOX: positions per band of the slices spread by value (KMX_SPLIT) against the m = 3 probe.
for t in ${KMX_SPLIT_SWEEP:-24576 28672 20480 16384}; do
  echo "== KMX_SPLIT=$t"
  KMX_SPLIT=$t python -c "from kmer_index_amd import build; build.build(force=True)" || exit 1
  timeout -k 10 200 python tools/probe_prefix.py 3 2>&1 | grep "^m=" | cut -c1-330
done
python -c "from kmer_index_amd import build; build.build(force=True)"
