#!/bin/bash
# Runs ON THE GPU BOX: SQ counters (issue / wait / LDS) for the kernels of one bench config — the evidence behind the
# "bound" column of DESIGN.md section 5 for the kernels that are not HBM bound (k_validate on config 3).
# Usage: bash tools/profile_sq.sh <tag> <config>   -> gpurun_out/sq_<tag>_cfg<config>/{pass1,pass2}/...
set -o pipefail
tag=${1:-cur}
cfg=${2:-3}
out=gpurun_out/sq_${tag}_cfg$cfg
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 -L > "$out/counters_available.txt" 2>&1
BENCH="bench.py --config $cfg --no-cpu-baseline --no-open-compare --no-two-streams --no-other-configs --no-host-api --steps 4 --warmup 1"
# any other python command instead of the bench (a probe; its name goes where the config number would):
#   KMX_SQ_CMD="tools/probe_sweep.py dna4_k=10" bash tools/profile_sq.sh <tag> <name>
if [ -n "$KMX_SQ_CMD" ]; then BENCH="$KMX_SQ_CMD"; fi
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES -d "$out/pass1" -o p1 --output-format csv -- python3 $BENCH > "$out/pass1.log" 2>&1
echo "pass1 rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD -d "$out/pass2" -o p2 --output-format csv -- python3 $BENCH > "$out/pass2.log" 2>&1
echo "pass2 rc=$?"
ls "$out"/pass1 "$out"/pass2 2>/dev/null
