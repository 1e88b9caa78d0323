#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the bench line plus the three rocprofv3 passes the roofline numbers come from.
# Usage: bash tools/profile_round.sh <tag> [config]   -> gpurun_out/prof_<tag>[_cfgN]/{bench.json,kt/,fetch/,write/}
#        (config = 2 (default, the metric's workload) | 3 | 4 | 5)
# Afterwards, in the container: python tools/summarise_profiles.py <tag> [--config N]  (copies the summaries into profiles/).
set -e -o pipefail
tag=${1:-cur}
cfg=${2:-2}
out=gpurun_out/prof_$tag
CFG=""
if [ "$cfg" != "2" ]; then out=${out}_cfg$cfg; CFG="--config $cfg"; fi
mkdir -p "$out"
export TMPDIR=/tmp
BENCH="bench.py $CFG --no-cpu-baseline --no-open-compare --no-two-streams --no-other-configs --no-host-api --steps 8 --warmup 2"
timeout -k 10 500 python3 bench.py $CFG > "$out/bench.json" 2> "$out/bench.err"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt --output-format csv -- python3 $BENCH > "$out/kt.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o fetch --output-format csv -- python3 $BENCH > "$out/fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o write --output-format csv -- python3 $BENCH > "$out/write.log" 2>&1
rm -f "$out"/kt/kt_kernel_trace.csv      # large; the stats summary is what is kept
cat "$out/bench.json"
