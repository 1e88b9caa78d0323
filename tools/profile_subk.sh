#!/bin/bash
# Runs ON THE GPU BOX: kernel trace and HBM counters (separate passes) of the sub-k probe — the evidence behind the sub-k table of
# DESIGN.md section 6.  Usage: bash tools/profile_subk.sh <tag>   -> gpurun_out/subk_<tag>/{kt,fetch,write}/...
# Afterwards, in the container: python tools/summarise_subk.py <tag>
set -o pipefail
tag=${1:-cur}
out=gpurun_out/subk_$tag
mkdir -p "$out"
export TMPDIR=/tmp
CMD="tools/probe_prefix.py 7 6 5 3"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt --output-format csv -- python3 $CMD > "$out/kt.log" 2>&1; echo "kt rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o fetch --output-format csv -- python3 $CMD > "$out/fetch.log" 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o write --output-format csv -- python3 $CMD > "$out/write.log" 2>&1; echo "write rc=$?"
rm -f "$out"/kt/kt_kernel_trace.csv
grep "^m=" "$out/kt.log" | cut -c1-120
