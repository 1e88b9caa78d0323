#!/bin/bash
# Runs ON THE GPU BOX: every informational probe DESIGN.md quotes, logs under gpurun_out/probes_<tag>/ (copy what is quoted
# into profiles/).  Usage: bash tools/run_probes.sh <tag>
tag=${1:-cur}
out=gpurun_out/probes_$tag
mkdir -p "$out"
run() { name=$1; shift; echo "== $name"; timeout -k 10 600 "$@" > "$out/$name.log" 2>&1; echo "rc=$?"; tail -3 "$out/$name.log"; }
run latency_py python tools/probe_latency.py
run latency_cpp python tools/probe_latency_cpp.py
KMX_NO_SMALL=1 run latency_cpp_general python tools/probe_latency_cpp.py
run host_api python tools/probe_host_api.py
run largek python tools/probe_largek.py
run skew python tools/probe_skew.py
run prefix python tools/probe_prefix.py
KMX_PREFIX_LEVELS=-1 run prefix_no_levels python tools/probe_prefix.py 9 8 6 5 3
KMX_PROBE=aa20 run prefix_aa20 python tools/probe_prefix.py
SKIP_HOST=1 run build python tools/probe_build.py
KMX_SWEEP_N=100000000 run sweep_1e8 python tools/probe_sweep.py
