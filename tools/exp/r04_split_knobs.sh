#!/bin/bash
# Runs ON THE GPU BOX: the tile size / threads of k_prefix_split_scatter against the m = 3 sub-k probe (rebuilds libkmx.so per case,
# leaves the default build behind).  Usage: bash tools/exp/r04_split_knobs.sh > gpurun_out/<dir>/split_knobs.log
for cfg in "8192 512" "4096 256" "4096 512" "8192 1024" "2048 256"; do
  set -- $cfg
  echo "== KMX_SPLIT_TILE=$1 KMX_SPLIT_SCATTER_THREADS=$2"
  KMX_SPLIT_TILE=$1 KMX_SPLIT_SCATTER_THREADS=$2 python -c "from kmer_index_amd import build; build.build(force=True)" || exit 1
  timeout -k 10 200 python tools/probe_prefix.py 3 2>&1 | grep "^m="
done
python -c "from kmer_index_amd import build; build.build(force=True)"
