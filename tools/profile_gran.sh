#!/bin/bash
# Runs ON THE GPU BOX: the random-read granularity microbenchmark (tools/micro/gather_gran.hip) on its own, then under
# rocprofv3 --pmc FETCH_SIZE — what a random 4 / 16 / 32 / 64 / 128-byte read costs, and what the gfx950 FETCH_SIZE counter
# says about it (the x2 correction of MI355X_MICROARCH.md is calibrated on wide streaming reads only).
# Usage: bash tools/profile_gran.sh <tag>   -> gpurun_out/gran_<tag>/{timing.log,fetch/}
# Afterwards, in the container: python tools/summarise_gran.py <tag>
set -o pipefail
tag=${1:-cur}
out=gpurun_out/gran_$tag
mkdir -p "$out"
export TMPDIR=/tmp
bin=tools/micro/gather_gran.bin
if [ ! -x "$bin" ]; then hipcc --offload-arch=gfx950 -O3 -o "$bin" tools/micro/gather_gran.hip || exit 1; fi
timeout -k 10 300 "$bin" 4096 12500000 > "$out/timing.log" 2>&1
echo "timing rc=$?"
timeout -k 10 300 "$bin" 100 12500000 > "$out/timing_100MiB.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o fetch --output-format csv -- "$bin" 4096 12500000 > "$out/fetch.log" 2>&1
echo "fetch rc=$?"
cat "$out/timing.log"
